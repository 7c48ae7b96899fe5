// Micro-experiment for pass 1 of the panel residual: does it matter WHICH lanes of a wave store?  A wave streams 64-entry chunks
// (8 B per lane) and leaves `per` partial sums per chunk in consecutive slots.
//   mode 0: the LAST lane of every piece stores (sparse exec mask, what ehyb_pb_scale_kernel does)
//   mode 1: the sums are first moved to lanes 0 .. per-1 (ds_bpermute), those lanes store (dense exec mask, same addresses)
//   mode 2: no stores at all (the bare stream)
// hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const double* __restrict__ val, double* __restrict__ out, long chunks, int piece)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    const int per = 64 / piece;
    double acc = 0.0;
    for (long c0 = wave * 8; c0 < chunks; c0 += nwaves * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c0 + j < chunks ? val[(c0 + j) * 64 + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (c0 + j >= chunks) break;
            double s = v[j] * 1.0000001;
            // piece sum by a couple of DPP-like shuffles (stand-in for the segmented scan)
            s += __shfl_up(s, 1, 64);
            s += __shfl_up(s, 2, 64);
            const long base = (c0 + j) * per;
            if (MODE == 0) {
                if ((lane % piece) == piece - 1) out[base + lane / piece] = s;
            } else if (MODE == 1) {
                const int src = lane * piece + piece - 1;                     // lane l < per takes the sum of piece l
                const int lo = __builtin_amdgcn_ds_bpermute((src & 63) << 2, __double2loint(s));
                const int hi = __builtin_amdgcn_ds_bpermute((src & 63) << 2, __double2hiint(s));
                if (lane < per) out[base + lane] = __hiloint2double(hi, lo);
            } else {
                acc += s;
            }
        }
    }
    if (MODE == 2 && acc == 123.456) out[0] = acc;
}

int main()
{
    const long entries = 33l << 20, chunks = entries / 64;
    double *val, *out;
    hipMalloc(&val, entries * 8);
    hipMalloc(&out, entries * 8);
    hipMemset(val, 0, entries * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int piece : {2, 4, 8}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(a, 0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(1024), 0, 0, val, out, chunks, piece);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(1024), 0, 0, val, out, chunks, piece);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(1024), 0, 0, val, out, chunks, piece);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("piece %d lanes (%2d stores per chunk, %5.1f MB written)  mode %d (%s): %7.1f us\n", piece, 64 / piece, chunks * (64 / piece) * 8 / 1e6, mode,
                   mode == 0 ? "last lane of each piece stores" : mode == 1 ? "compacted to lanes 0..n-1   " : "no stores                   ", best * 1e3);
        }
    }
    return 0;
}
