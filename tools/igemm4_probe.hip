// Stand-alone probe of the phase-interleaved conv kernel (pgtformer_amd/csrc/igemm4.hip): times one conv shape with
// HIP events and, with -DPGT_PROBE=32|..., prints the per-workgroup segment lengths measured with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I pgtformer_amd/csrc -DPGT_PROBE=<bits> \
//         tools/igemm4_probe.hip -o gpurun_out/probe_<bits>
//   probe N H W Cin Cout k [iters]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <utility>

#ifdef PGT_PROBE_V5
#include "../pgtformer_amd/csrc/igemm5.hip"
#define PROBE_LAUNCH(p, bn) pgt_igemm5_launch(p, 0)
#else
#include "../pgtformer_amd/csrc/igemm4.hip"
#define PROBE_LAUNCH(p, bn) pgt_igemm4_launch(p, bn, 0)
#endif

void pgt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

__global__ void fill_bf16(bf16_t* p, long n, unsigned seed, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i].v = f2bf(((h & 0xffff) / 32768.f - 1.f) * scale);
    }
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: probe N H W Cin Cout k [iters] [bn=256|128] [ups=0|1]\n"); return 2; }
    const int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]), k = atoi(argv[6]);
    const int iters = argc > 7 ? atoi(argv[7]) : 5;
    const int bn = argc > 8 ? atoi(argv[8]) : 256, ups = argc > 9 ? atoi(argv[9]) : 0;
    ConvP p{};
    const long nx = (long)N * H * W * Cin, nw = (long)Cout * k * k * Cin, ny = (long)N * H * W * Cout * 4;
    bf16_t *x, *w, *y;
    float* bias;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&bias, Cout * 4);
    fill_bf16<<<1024, 256>>>(x, nx, 1u, 1.f);
    fill_bf16<<<1024, 256>>>(w, nw, 7u, 0.05f);
    hipMemset(bias, 0, Cout * 4);
    p.x = (const char*)x; p.w = (const char*)w; p.bias = bias; p.y = (char*)y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.ldx = Cin; p.KH = p.KW = k; p.stride = 1; p.pad_t = p.pad_l = k / 2;
    p.ups = ups; p.Ho = ups ? 2 * H : H; p.Wo = ups ? 2 * W : W; p.Cout = Cout; p.ldy = Cout; p.vec_epi = 1;
    p.M = N * p.Ho * p.Wo; p.K = k * k * Cin; p.nw = Cout;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    PROBE_LAUNCH(&p, bn);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) PROBE_LAUNCH(&p, bn);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, flops = 2.0 * p.M * Cout * p.K;
    printf("PROBE=%d  %dx%dx%dx%d -> %d k%d : %.1f us  %.1f TFLOP/s  (%s)\n", PGT_PROBE, N, H, W, Cin, Cout, k, us,
           flops / us / 1e6, hipGetErrorString(hipGetLastError()));
#if PGT_PROBE & 32
    const int bm = bn == 256 ? 256 : 512;
    const int nb = ((p.M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    std::vector<unsigned long long> ts(4096 * 8);
    hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_pgt_probe_ts), ts.size() * 8);
    double s[3] = {0, 0, 0}, s4 = 0, s5 = 0;
    const int cnt = nb < 4096 ? nb : 4096;
    for (int b = 0; b < cnt; ++b)
        for (int j = 0; j < 3; ++j) s[j] += (double)(ts[b * 8 + j + 1] - ts[b * 8 + j]);
    for (int b = 0; b < cnt; ++b) { s4 += (double)(ts[b * 8 + 4] - ts[b * 8]); s5 += (double)(ts[b * 8 + 5] - ts[b * 8 + 4]); }
    printf("  setup split: address arithmetic %.0f  prologue issue + first wait %.0f  barrier %.0f\n", s4 / cnt, s5 / cnt,
           (s[0] - s4 - s5) / cnt);
    {   // gaps between consecutive workgroups of one CU, and the counter rate against the event time
        std::vector<std::pair<unsigned long long, int>> order;
        for (int b = 0; b < cnt; ++b) order.push_back({((ts[b * 8 + 6] & 0xf0000ffffull) >> 8 << 8) | 0, b});
        // key = (xcc, se, sh, cu): HW_ID bits [15:8], XCC_ID bits [3:0] of the high word
        std::vector<std::vector<int>> percu(8 * 256);
        for (int b = 0; b < cnt; ++b) {
            const unsigned hw = (unsigned)ts[b * 8 + 6], xcc = (unsigned)(ts[b * 8 + 6] >> 32) & 0xf;
            percu[(xcc & 7) * 256 + ((hw >> 8) & 0xff)].push_back(b);
        }
        double gap = 0, span = 0; int ngap = 0, ncu = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (auto& v : percu) {
            if (v.empty()) continue;
            ++ncu;
            std::sort(v.begin(), v.end(), [&](int a, int b) { return ts[a * 8] < ts[b * 8]; });
            for (size_t i = 1; i < v.size(); ++i) { gap += (double)(ts[v[i] * 8] - ts[v[i - 1] * 8 + 3]); ++ngap; }
            span += (double)(ts[v.back() * 8 + 3] - ts[v.front() * 8]);
            if (ts[v.front() * 8] < tmin) tmin = ts[v.front() * 8];
            if (ts[v.back() * 8 + 3] > tmax) tmax = ts[v.back() * 8 + 3];
        }
        printf("  %d CUs used, %.1f workgroups per CU; gap between consecutive workgroups of a CU %.0f ticks (avg of %d); "
               "per-CU busy span %.0f ticks; event time %.1f us -> %.3f ticks per ns\n",
               ncu, (double)cnt / ncu, ngap ? gap / ngap : 0.0, ngap, span / ncu, us, span / ncu / (us * 1e3));
    }
    printf("  per workgroup (s_memtime ticks, avg of %d): setup+prologue %.0f  main loop %.0f (%.0f per K tile, %.0f per phase)  epilogue %.0f\n",
           cnt, s[0] / cnt, s[1] / cnt, s[1] / cnt / (p.K / 64), s[1] / cnt / (p.K / 64) / 4, s[2] / cnt);
#endif
    return 0;
}
