// Standalone timing + correctness bench of the dense attention kernels (da::launch_attn_dense), no Python:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffassemble_amd/csrc tools/attn_bench.hip -Ldiffassemble_amd/lib -ldiffassemble_hip \
//         -Wl,-rpath,'$ORIGIN/../../diffassemble_amd/lib' -o tools/bin/attn_bench
//   tools/bin/attn_bench <G> <n> <C> <fold 0|1> [iters] [check 0|1] [nodiag 0|1] [spread] [dual 0|1]
// Q / K / V in the head-major padded layout the projections write ([H][n_pad][C], every graph in a 64-row-aligned slot);
// fold = 1: 32-wide value heads, per-head normalised outputs [H][N][32] (the folded last layer of the 2D arch);
// fold = 0: C-wide values, out = GELU(attn + skip) [N][H*C] (a hidden layer).  check = 1 compares with a plain fp32
// kernel (one thread per (query, head)) on the same bf16 inputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "da_common.h"
#include "da_internal.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using da::bf16_t;

__global__ void k_ref(const bf16_t *Q, const bf16_t *K, const bf16_t *V, const bf16_t *S, int G, int n, int npg, int n_pad, int H, int C, int CV,
                      int fold, int nodiag, float qk_scale_log2, float *out) {
    // one thread per (graph, query, head); out fp32: fold -> [H][N][CV] normalised; else [N][H*C] = gelu(attn + skip)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * n * H) return;
    const int h = idx % H, q = (idx / H) % n, g = idx / (H * n);
    const bf16_t *qr = Q + ((size_t)h * n_pad + (size_t)g * npg + q) * C;
    float m = -INFINITY, l = 0.f, acc[144];
    for (int c = 0; c < CV; ++c) acc[c] = 0.f;
    for (int j = 0; j < n; ++j) {
        if (nodiag && j == q) continue;
        const bf16_t *kr = K + ((size_t)h * n_pad + (size_t)g * npg + j) * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += da::bf2f(qr[c]) * da::bf2f(kr[c]);
        s *= qk_scale_log2;
        const float mn = fmaxf(m, s), corr = exp2f(m - mn), p = exp2f(s - mn);
        l = l * corr + p;
        const bf16_t *vr = V + ((size_t)h * n_pad + (size_t)g * npg + j) * CV;
        for (int c = 0; c < CV; ++c) acc[c] = acc[c] * corr + p * da::bf2f(vr[c]);
        m = mn;
    }
    const float inv = l > 0.f ? 1.f / (l + 1e-16f) : 0.f;
    const size_t node = (size_t)g * n + q;
    if (fold) {
        for (int c = 0; c < CV; ++c) out[((size_t)h * G * n + node) * CV + c] = acc[c] * inv;
    } else {
        for (int c = 0; c < C; ++c) {
            const float v = acc[c] * inv + da::bf2f(S[node * H * C + h * C + c]);
            out[node * H * C + h * C + c] = 0.5f * v * (1.0f + erff(v * 0.70710678f));
        }
    }
}

static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 900, C = argc > 3 ? atoi(argv[3]) : 144;
    const int fold = argc > 4 ? atoi(argv[4]) : 1, iters = argc > 5 ? atoi(argv[5]) : 50, check = argc > 6 ? atoi(argv[6]) : 0;
    const int nodiag = argc > 7 ? atoi(argv[7]) : 0;
    const float spread = argc > 8 ? atof(argv[8]) : 1.0f;          // score spread: 1 = mild, 8 = sharp attention (rescale path)
    const int dual = argc > 9 ? atoi(argv[9]) : 0;                 // 1: da::launch_attn_dual directly; 2: da::launch_attn_dense with the pre-scaled flag (production dispatch); both: Q pre-scaled by log2(e) / sqrt(C)
    const int H = 8, CV = fold ? 32 : C, npg = (n + 63) / 64 * 64, n_pad = G * npg, N = G * n;
    std::vector<bf16_t> hq((size_t)H * n_pad * C), hk(hq.size()), hv((size_t)H * n_pad * CV), hs((size_t)N * H * C);
    unsigned seed = 12345;
    const float sc_l2 = 1.4426950408889634f / sqrtf((float)C);
    for (auto &x : hq) x = da::f2bf(frand(seed) * 2.f * spread * (dual ? sc_l2 : 1.0f));
    for (auto &x : hk) x = da::f2bf(frand(seed) * 2.f);
    for (auto &x : hv) x = da::f2bf(frand(seed) * 2.f);
    for (auto &x : hs) x = da::f2bf(frand(seed));
    std::vector<int32_t> gp(G + 1), pp(G + 1);
    for (int g = 0; g <= G; ++g) { gp[g] = g * n; pp[g] = g * npg; }
    bf16_t *dq, *dk, *dv, *ds, *dout; int32_t *dgp, *dpp; float *dref;
    const size_t out_elems = fold ? (size_t)H * N * CV : (size_t)N * H * C;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dk, hk.size() * 2)); CK(hipMalloc(&dv, hv.size() * 2)); CK(hipMalloc(&ds, hs.size() * 2));
    CK(hipMalloc(&dout, out_elems * 2)); CK(hipMalloc(&dgp, (G + 1) * 4)); CK(hipMalloc(&dpp, (G + 1) * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgp, gp.data(), (G + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpp, pp.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, out_elems * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    da::DenseLayout L; L.Q = dq; L.K = dk; L.Vt = dv; L.S = fold ? nullptr : ds; L.n_pad = n_pad; L.q_prescaled = dual ? 1 : 0;
    da::DenseFold fo; fo.cv = 32; fo.out = dout; fo.n_rows = N;
    auto run = [&]() {
        if (dual == 1) return da::launch_attn_dual(L, H, C, G, n, dgp, dpp, nodiag, (fold || getenv("ACT_NONE")) ? DA_ACT_NONE : DA_ACT_GELU, fold ? nullptr : dout, fold ? &fo : nullptr, st);
        return da::launch_attn_dense(DA_PREC_BF16, L, H, C, G, n, dgp, dpp, nodiag, nullptr, (fold || getenv("ACT_NONE")) ? DA_ACT_NONE : DA_ACT_GELU, fold ? nullptr : dout, st,
                                     nullptr, fold ? &fo : nullptr);
    };
    int rc = run();
    if (rc) { printf("launch rc %d: %s\n", rc, da_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    if (check) {
        CK(hipMalloc(&dref, out_elems * 4));
        const float sc = dual ? 1.0f : sc_l2;
        k_ref<<<(G * n * H + 255) / 256, 256, 0, st>>>(dq, dk, dv, ds, G, n, npg, n_pad, H, C, CV, fold, nodiag, sc, dref);
        CK(hipStreamSynchronize(st));
        std::vector<float> href(out_elems); std::vector<bf16_t> hout(out_elems);
        CK(hipMemcpy(href.data(), dref, out_elems * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hout.data(), dout, out_elems * 2, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; size_t bad = 0, nan = 0;
        for (size_t i = 0; i < out_elems; ++i) {
            const float a = da::bf2f(hout[i]), b = href[i];
            if (!(a == a)) { ++nan; continue; }
            maxerr = fmax(maxerr, fabs((double)a - b)); maxref = fmax(maxref, fabs((double)b));
            if (fabs((double)a - b) > 0.02 * fabs(b) + 0.01) {
                if (bad < 12 && fold) { const size_t row = i / CV; printf("  bad: h %zu node %zu (q %zu) c %zu got %.4f ref %.4f\n", row / N, row % N, (row % N) % n, i % CV, a, b); }
                ++bad;
            }
        }
        if (fold && bad) {          // bad elements per 32-query slab of graph 0, head 0
            std::vector<int> hist((n + 31) / 32, 0);
            for (size_t i = 0; i < out_elems; ++i) {
                const float a = da::bf2f(hout[i]), b = href[i];
                if (fabs((double)a - b) > 0.02 * fabs(b) + 0.01) hist[((i / CV) % N) % n / 32]++;
            }
            printf("  bad per slab:"); for (int v : hist) printf(" %d", v); printf("\n");
        }
        printf("check: max|err| %.4g  max|ref| %.4g  norm-wise %.3g  elements off by > 2%%+0.01: %zu  nan: %zu of %zu\n", maxerr, maxref, maxerr / maxref, bad, nan, out_elems);
    }
    if (getenv("PROBE")) {          // DA_DUAL_PROBE build of the library: cycle breakdown of wave 0 of every workgroup
        const int nwg = 8 * G * 8;
        unsigned long long *dprof; CK(hipMalloc(&dprof, nwg * 4 * 8)); CK(hipMemset(dprof, 0, nwg * 4 * 8));
        char buf[64]; snprintf(buf, sizeof buf, "%llu", (unsigned long long)dprof); setenv("DA_DUAL_PROF_PTR", buf, 1);
        run(); run(); CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp(nwg * 4); CK(hipMemcpy(hp.data(), dprof, nwg * 4 * 8, hipMemcpyDeviceToHost));
        double tot[4] = {0, 0, 0, 0}; int cnt = 0;
        for (int w = 0; w < nwg; ++w) if (hp[4 * w]) { for (int k = 0; k < 4; ++k) tot[k] += hp[4 * w + k]; ++cnt; }
        printf("probe (wave 0, %d workgroups, 100 MHz ticks): total %.0f  sync %.0f  region1 %.0f  region2+pv1 %.0f\n", cnt, tot[0] / cnt, tot[1] / cnt, tot[2] / cnt, tot[3] / cnt);
        unsetenv("DA_DUAL_PROF_PTR");
    }
    if (getenv("PROBE2")) {         // DA_OPT_PROBE build of the library (da_attn_opt.hip): cycle breakdown of wave 0 of every workgroup
        const int nwg = 8 * G * 8;
        unsigned long long *dprof; CK(hipMalloc(&dprof, nwg * 16 * 8)); CK(hipMemset(dprof, 0, nwg * 16 * 8));
        char buf[64]; snprintf(buf, sizeof buf, "%llu", (unsigned long long)dprof); setenv("DA_OPT_PROF_PTR", buf, 1);
        for (int i = 0; i < 4; ++i) run();
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp(nwg * 16); CK(hipMemcpy(hp.data(), dprof, nwg * 16 * 8, hipMemcpyDeviceToHost));
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int cnt = 0;
        unsigned long long w0 = ~0ull, w1 = 0;
        std::vector<double> dur, start, endt;
        for (int w = 0; w < nwg; ++w) if (hp[16 * w + 7]) { w0 = std::min(w0, hp[16 * w + 8]); w1 = std::max(w1, hp[16 * w + 9]); }
        for (int w = 0; w < nwg; ++w) if (hp[16 * w + 7] && hp[16 * w + 6] > 8) {
            for (int k = 0; k < 8; ++k) tot[k] += hp[16 * w + k];
            ++cnt;
            dur.push_back((double)(hp[16 * w + 9] - hp[16 * w + 8]) * 0.01); start.push_back((double)(hp[16 * w + 8] - w0) * 0.01); endt.push_back((double)(hp[16 * w + 9] - w0) * 0.01);
        }
        printf("probe2 (wave 0 of %d full workgroups, cycles): kernel %.0f | tile wait+barrier %.0f | DMA issue %.0f | K-fragment reads %.0f | QK+softmax+PV %.0f | epilogue %.0f | blocks %.1f\n",
               cnt, tot[0] / cnt, tot[1] / cnt, tot[2] / cnt, tot[3] / cnt, tot[4] / cnt, tot[5] / cnt, tot[6] / cnt);
        printf("   per block: wait %.0f  dma %.0f  kread %.0f  math %.0f   (prologue+rest %.0f)\n", tot[1] / tot[6], tot[2] / tot[6], tot[3] / tot[6], tot[4] / tot[6],
               (tot[0] - tot[1] - tot[2] - tot[3] - tot[4] - tot[5]) / cnt);
        { double p1 = 0, p2 = 0, p3 = 0; int c2 = 0; for (int w = 0; w < nwg; ++w) if (hp[16 * w + 7] && hp[16 * w + 6] > 8) { p1 += hp[16 * w + 12]; p2 += hp[16 * w + 13]; p3 += hp[16 * w + 14]; ++c2; }
          printf("   before the loop: kernel entry -> graph table read %.0f | -> Q fragments landed %.0f | -> first tile requested %.0f cycles\n", p1 / c2, p2 / c2, p3 / c2); }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        printf("   wall clock (us, s_memrealtime): first start -> last end %.2f | workgroup duration min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f | start p50 %.2f p90 %.2f max %.2f | end p10 %.2f p50 %.2f\n",
               (double)(w1 - w0) * 0.01, pct(dur, 0), pct(dur, 0.1), pct(dur, 0.5), pct(dur, 0.9), pct(dur, 1), pct(start, 0.5), pct(start, 0.9), pct(start, 1), pct(endt, 0.1), pct(endt, 0.5));
        // occupancy over time: workgroups in flight at 10 equally spaced instants
        printf("   workgroups in flight at t = k/10 of the span:");
        for (int k = 1; k < 10; ++k) { const double t = (double)(w1 - w0) * 0.01 * k / 10; int c = 0; for (size_t x = 0; x < start.size(); ++x) c += start[x] <= t && endt[x] > t; printf(" %d", c); }
        printf("\n");
        unsetenv("DA_OPT_PROF_PTR");
    }
    if (getenv("PROBE3")) {         // DA_OPT_PROBE build, k_attn_res: per-wave stamps [entry, DMA issued, pass 0 end, slab 0 end, pass 1 end, slab 1 end, first barrier passed, exit]
        const int nwv = 16, nwg = G * 8;
        unsigned long long *dprof; CK(hipMalloc(&dprof, (size_t)nwg * nwv * 64)); CK(hipMemset(dprof, 0, (size_t)nwg * nwv * 64));
        char buf[64]; snprintf(buf, sizeof buf, "%llu", (unsigned long long)dprof); setenv("DA_OPT_PROF_PTR", buf, 1);
        for (int i = 0; i < 3; ++i) run();
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp((size_t)nwg * nwv * 8); CK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
        const char *nm[8] = {"entry", "DMA issued", "pass 0 end", "slab 0 end", "pass 1 end", "slab 1 end", "first barrier passed", "exit"};
        for (int grp = 0; grp < 2; ++grp) {
            double sum[8] = {0}; double mx[8] = {0}; int cnt = 0;
            for (int w = 0; w < nwg * nwv; ++w) {
                const unsigned long long *o = &hp[(size_t)w * 8];
                const bool two = (w % nwv) < 13;
                if (!o[0] || two != (grp == 0)) continue;
                ++cnt;
                for (int k = 1; k < 8; ++k) { const double d = o[k] ? (double)(o[k] - o[0]) : 0; sum[k] += d; mx[k] = std::max(mx[k], d); }
            }
            printf("probe3 %s (%d waves), cycles since entry, mean / max:", grp == 0 ? "waves with two slabs" : "waves with one slab", cnt);
            for (int k : {1, 6, 2, 3, 4, 5, 7}) printf("  %s %.0f / %.0f", nm[k], cnt ? sum[k] / cnt : 0, mx[k]);
            printf("\n");
        }
        unsetenv("DA_OPT_PROF_PTR");
    }
    for (int i = 0; i < 5; ++i) run();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, tot = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / iters); tot += ms / iters;
    }
    const double alg = (double)G * n * n * 4.0 * H * C, exe = (double)G * n * n * 2.0 * H * (C + CV);
    printf("G=%d n=%d C=%d CV=%d nodiag=%d spread=%g: %.1f us (best of 3 x %d; mean %.1f)  alg %.0f TF/s = %.3f of 2.5 PF  executed %.0f TF/s = %.3f\n", G, n, C, CV, nodiag, spread,
           best * 1e3, iters, tot / 3 * 1e3, alg / best / 1e9, alg / best / 1e9 / 2500, exe / best / 1e9, exe / best / 1e9 / 2500);
    return 0;
}
