// da_config: the product switches of the library (include/diffassemble_hip.h) -- environment defaults, run-time get / set.
#include <string.h>

#include <mutex>

#include "da_common.h"
#include "da_config.h"

namespace da {

static int env_or(const char *name, int dflt) { const char *e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; }

static da_config load_env() {
    da_config c;
    memset(&c, 0, sizeof c);
    c.struct_bytes = (int)sizeof(da_config);
    c.disable_mfma = env_or("DA_DISABLE_MFMA", 0);
    c.disable_dense = env_or("DA_DISABLE_DENSE", 0);
    c.disable_folds = env_or("DA_DISABLE_FOLDS", 0);
    c.attn_level = env_or("DA_ATTN_LEVEL", 2);
    c.xpanel = env_or("DA_ENABLE_XPANEL", -1);
    c.tail_next = env_or("DA_TAIL_NEXT", -1);
    c.pair_split = env_or("DA_PAIR_SPLIT", 1);
    c.train_attn = env_or("DA_TRAIN_ATTN", 2);
    c.train_side_streams = env_or("DA_TRAIN_SIDE_STREAMS", 3);
    return c;
}

da_config &cfg() {
    static da_config c = load_env();
    return c;
}

}  // namespace da

extern "C" {

int da_config_get(da_config *out) {
    DA_REQUIRE(out, "da_config_get: null argument");
    *out = da::cfg();
    return 0;
}

int da_config_set(const da_config *in) {
    DA_REQUIRE(in && in->struct_bytes == (int)sizeof(da_config), "da_config_set: struct_bytes %d != %d (header / library mismatch)",
               in ? in->struct_bytes : -1, (int)sizeof(da_config));
    da::cfg() = *in;
    return 0;
}

int da_build_flags(void) {
#ifdef DA_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
