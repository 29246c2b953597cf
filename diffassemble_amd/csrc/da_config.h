// The library's switches in ONE place.
//
// PRODUCT switches: the fields of `da_config` (include/diffassemble_hip.h), initialised once per process from their documented
// environment variables and settable at run time through da_config_set (in-process A/Bs: capture one graph per setting and replay
// them interleaved).  Call sites read `da::cfg().field` -- no other getenv on the product path.
//
// EXPERIMENT switches (A/B variants that lost, timing ablations, probes): DA_XENV("NAME", default) is the CONSTANT `default` in the
// product build; only a library compiled with -DDA_EXPERIMENTS (DA_EXPERIMENTS=1 python __graft_entry__.py -> lib_exp/) reads the
// environment for them, and only that build contains the kernels they select.
#pragma once
#include <stdlib.h>

#include "../../include/diffassemble_hip.h"

namespace da {

da_config &cfg();

// disable_folds bits
enum { DA_FOLD_MLP2 = 1, DA_FOLD_LAST = 2, DA_FOLD_QSCALE = 4, DA_FOLD_DDIM = 8, DA_FOLD_TAIL = 16, DA_FOLD_HYBRID_OVERLAP = 32 };

#ifdef DA_EXPERIMENTS
inline int xenv_read(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
// cached per call site (the value a process starts with), like the statics these replace
#define DA_XENV(name, dflt) ([]() -> int { static const int v_ = da::xenv_read(name, dflt); return v_; }())
// read on every call (probe pointers, per-launch debug masks)
#define DA_XENV_LIVE(name) getenv(name)
#define DA_XENV_SET(name) (getenv(name) != nullptr)
#else
#define DA_XENV(name, dflt) (dflt)
#define DA_XENV_LIVE(name) ((const char *)nullptr)
#define DA_XENV_SET(name) (false)
#endif

}  // namespace da
