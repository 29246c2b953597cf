"""Drop-in counterpart of the reference's ``puzzle_diff/model`` package (same module and
class names, constructor kwargs and state-dict keys)."""
