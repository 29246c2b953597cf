"""Import-name placeholder for puzzle_diff/model/spatial_diffusion_discrete.py: the reference drivers import this
module (train_script.py:19,25-26; viz_script.py:19-20) but only use it behind non-default
flags.  Experimental variant, outside the accelerated hot path."""


class GNN_Diffusion:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("spatial_diffusion_discrete.GNN_Diffusion is out of scope (SURVEY.md section 2, #12)")
