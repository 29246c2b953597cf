// Co-residency probe: does the four-wave projection kernel of da_gemm_thin.hip run INSIDE the CUs that the
// K / V-resident hidden-layer attention kernel occupies, and what does each pay for it?
//   build: tools/build_tools.sh (EXPERIMENTS build of the library);   run: tools/bin/corun_probe [G=32] [n=900] [iters=200]
// 1. bit-identity: thin vs the default W-in-registers kernels on the three projection shapes of the 2D denoiser.
// 2. timing: attention alone, each projection kernel alone, and both back to back on TWO streams at once (wall time per pair of launches).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "da_common.h"
#include "da_internal.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using da::bf16_t;
static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Proj {          // one fused projection: x [M, K] -> Q | K | V (| skip) head-major at the padded slots
    int M, K, Nout, HC, C, Cv, n_pad;
    bf16_t *x, *w, *q, *k, *v, *s; float *b; int32_t *row_map;
    size_t qn, vn, sn;
};

static Proj make_proj(int G, int n, int K, int C, int Cv, unsigned seed) {
    Proj p; const int H = 8, npg = (n + 63) / 64 * 64;
    p.M = G * n; p.K = K; p.HC = H * C; p.C = C; p.Cv = Cv; p.n_pad = G * npg;
    p.Nout = Cv > 0 ? 2 * p.HC + H * Cv : 4 * p.HC;
    std::vector<bf16_t> hx((size_t)p.M * K), hw((size_t)p.Nout * K); std::vector<float> hb(p.Nout); std::vector<int32_t> rm(p.M);
    for (auto &e : hx) e = da::f2bf(frand(seed) * 2.f);
    for (auto &e : hw) e = da::f2bf(frand(seed) * 0.25f);
    for (auto &e : hb) e = frand(seed);
    for (int g = 0; g < G; ++g) for (int i = 0; i < n; ++i) rm[g * n + i] = g * npg + i;
    p.qn = (size_t)H * p.n_pad * C; p.vn = (size_t)H * p.n_pad * (Cv > 0 ? Cv : C); p.sn = Cv > 0 ? 0 : (size_t)p.M * p.HC;
    CK(hipMalloc(&p.x, hx.size() * 2)); CK(hipMalloc(&p.w, hw.size() * 2)); CK(hipMalloc(&p.b, hb.size() * 4)); CK(hipMalloc(&p.row_map, rm.size() * 4));
    CK(hipMalloc(&p.q, p.qn * 2)); CK(hipMalloc(&p.k, p.qn * 2)); CK(hipMalloc(&p.v, p.vn * 2)); p.s = nullptr;
    if (p.sn) CK(hipMalloc(&p.s, p.sn * 2));
    CK(hipMemcpy(p.x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(p.w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p.row_map, rm.data(), rm.size() * 4, hipMemcpyHostToDevice));
    return p;
}
static int run_proj(const Proj &p, hipStream_t st) {
    da::QkvScatter qs; qs.HC = p.HC; qs.C = p.C; qs.Cv = p.Cv; qs.n_pad = p.n_pad; qs.row_map = p.row_map; qs.Q = p.q; qs.K = p.k; qs.Vt = p.v; qs.S = p.s;
    return da::launch_gemm_mfma(DA_PREC_BF16, p.M, p.K, p.Nout, p.x, p.K, p.w, p.b, DA_ACT_NONE, nullptr, nullptr, 0, &qs, st);
}
static std::vector<bf16_t> grab(const Proj &p) {
    std::vector<bf16_t> h(2 * p.qn + p.vn + p.sn);
    CK(hipMemcpy(h.data(), p.q, p.qn * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data() + p.qn, p.k, p.qn * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h.data() + 2 * p.qn, p.v, p.vn * 2, hipMemcpyDeviceToHost));
    if (p.sn) CK(hipMemcpy(h.data() + 2 * p.qn + p.vn, p.s, p.sn * 2, hipMemcpyDeviceToHost));
    return h;
}
static void wipe(const Proj &p) { CK(hipDeviceSynchronize()); CK(hipMemset(p.q, 0xff, p.qn * 2)); CK(hipMemset(p.k, 0xff, p.qn * 2)); CK(hipMemset(p.v, 0xff, p.vn * 2)); if (p.sn) CK(hipMemset(p.s, 0xff, p.sn * 2)); CK(hipDeviceSynchronize()); }   // (the kernels run on non-blocking streams: the memsets of the null stream must have landed)

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 32, n = argc > 2 ? atoi(argv[2]) : 900, iters = argc > 3 ? atoi(argv[3]) : 200;
    const int H = 8, C = 32, npg = (n + 63) / 64 * 64, n_pad = G * npg, N = G * n;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // ---- 1. bit-identity on the three shapes (conv 0: K 128 x 1024; hidden: K 256 x 1024; folded last: K 256 x 2560, C 144, Cv 32)
    Proj shapes[3] = {make_proj(G, n, 128, 32, 0, 1), make_proj(G, n, 256, 32, 0, 2), make_proj(G, n, 256, 144, 32, 3)};
    const char *names[3] = {"conv0 K128 N1024", "hidden K256 N1024", "last K256 N2560"};
    for (int i = 0; i < 3; ++i) {
        wipe(shapes[i]); da::gemm_thin_set(0);
        if (run_proj(shapes[i], sa)) { printf("ref launch failed: %s\n", da_last_error()); return 1; }
        CK(hipStreamSynchronize(sa)); const auto ref = grab(shapes[i]);
        wipe(shapes[i]); da::gemm_thin_set(1);
        if (run_proj(shapes[i], sa)) { printf("thin launch failed: %s\n", da_last_error()); return 1; }
        CK(hipStreamSynchronize(sa)); const auto got = grab(shapes[i]);
        size_t diff = 0, untouched = 0;
        for (size_t e = 0; e < ref.size(); ++e) { diff += ref[e] != got[e]; untouched += (got[e] == 0xffff && ref[e] == 0xffff); }
        printf("bit-identity %-18s: %zu of %zu elements differ (%zu padded slots untouched by both)\n", names[i], diff, ref.size(), untouched);
        if (diff) {          // where: block (Q / K / V / skip), head, slot, column; how far apart
            const Proj &p = shapes[i]; size_t shown = 0, blk[4] = {0, 0, 0, 0}, unwritten = 0; double maxd = 0;
            std::vector<size_t> percol(160, 0);
            for (size_t e = 0; e < ref.size(); ++e) {
                if (ref[e] == got[e]) continue;
                const int b = e < p.qn ? 0 : (e < 2 * p.qn ? 1 : (e < 2 * p.qn + p.vn ? 2 : 3));
                const size_t o = e - (b == 0 ? 0 : (b == 1 ? p.qn : (b == 2 ? 2 * p.qn : 2 * p.qn + p.vn)));
                const int cw = b == 2 && p.Cv > 0 ? p.Cv : (b == 3 ? p.HC : p.C);
                ++blk[b]; ++percol[(o % cw) % 160]; unwritten += got[e] == 0xffff;
                maxd = fmax(maxd, fabs((double)da::bf2f(ref[e]) - da::bf2f(got[e])));
                if (shown++ < 6) printf("    block %d head/row %zu col %zu: ref %.5f thin %.5f (raw %04x %04x)\n", b, o / cw, o % cw, da::bf2f(ref[e]), da::bf2f(got[e]), ref[e], got[e]);
            }
            printf("    per block Q %zu K %zu V %zu S %zu; left unwritten by thin %zu; max |diff| %.4g; per column-in-head:", blk[0], blk[1], blk[2], blk[3], unwritten, maxd);
            for (int c = 0; c < (p.C > 32 ? p.C : 32); ++c) printf(" %zu", percol[c]);
            printf("\n");
        }
    }
    // ---- 2. attention inputs (a SECOND half Batch: its own Q / K / V / skip / out)
    std::vector<bf16_t> hq((size_t)H * n_pad * C), hk(hq.size()), hv(hq.size()), hs((size_t)N * H * C);
    unsigned seed = 99; const float sc_l2 = 1.4426950408889634f / sqrtf((float)C);
    for (auto &x : hq) x = da::f2bf(frand(seed) * 2.f * sc_l2);
    for (auto &x : hk) x = da::f2bf(frand(seed) * 2.f);
    for (auto &x : hv) x = da::f2bf(frand(seed) * 2.f);
    for (auto &x : hs) x = da::f2bf(frand(seed));
    std::vector<int32_t> gp(G + 1), pp(G + 1);
    for (int g = 0; g <= G; ++g) { gp[g] = g * n; pp[g] = g * npg; }
    bf16_t *dq, *dk, *dv, *ds, *dout; int32_t *dgp, *dpp;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dk, hk.size() * 2)); CK(hipMalloc(&dv, hv.size() * 2)); CK(hipMalloc(&ds, hs.size() * 2));
    CK(hipMalloc(&dout, hs.size() * 2)); CK(hipMalloc(&dgp, (G + 1) * 4)); CK(hipMalloc(&dpp, (G + 1) * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgp, gp.data(), (G + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpp, pp.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    da::DenseLayout L; L.Q = dq; L.K = dk; L.Vt = dv; L.S = ds; L.n_pad = n_pad; L.q_prescaled = 1;
    auto attn = [&](hipStream_t st) { return da::launch_attn_dense(DA_PREC_BF16, L, H, C, G, n, dgp, dpp, 0, nullptr, DA_ACT_GELU, dout, st, nullptr, nullptr); };
    // wall time per iteration of `fa` on stream sa and `fb` on stream sb, enqueued alternately (either may be null)
    auto wall = [&](auto fa, auto fb, bool has_a, bool has_b) {
        for (int w = 0; w < 10; ++w) { if (has_a) fa(sa); if (has_b) fb(sb); }
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int i = 0; i < iters; ++i) { if (has_a) fa(sa); if (has_b) fb(sb); }
        CK(hipDeviceSynchronize());
        return (now_us() - t0) / iters;
    };
    auto none = [&](hipStream_t) { return 0; };
    const double t_attn = wall(attn, none, true, false);
    printf("attention alone (k_attn_res, %d graphs x %d):            %7.1f us\n", G, n, t_attn);
    for (int i = 0; i < 3; ++i) {
        for (int thin = 0; thin < 2; ++thin) {
            da::gemm_thin_set(thin);
            auto pj = [&](hipStream_t st) { return run_proj(shapes[i], st); };
            const double t_p = wall(none, pj, false, true);
            const double t_both = wall(attn, pj, true, true);
            printf("%-18s %-5s: alone %6.1f us | beside attention: pair %6.1f us  (sum %6.1f, max %6.1f, hidden share of the projection %4.0f %%)\n", names[i],
                   thin ? "thin" : "wreg", t_p, t_both, t_attn + t_p, fmax(t_attn, t_p), 100.0 * (t_attn + t_p - t_both) / t_p);
        }
    }
    return 0;
}
