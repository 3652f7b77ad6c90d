// EXPERIMENT (not product): what do the dQ atomics of a single-pass attention backward cost on gfx950?
// Shape of the traffic: B*H (b,h) pairs, each with a [L, D] fp32 dQ buffer (L = 1280, D = 128: 655 KB, L2 resident on the pair's XCD);
// L / KB key blocks per pair, each walks the L / 32 query steps and adds a [32, D] fp32 partial per step (pre-reduced over the block's waves:
// mode "block"), or one partial per wave pair (mode "pair": 4 x the atoms).  All blocks of a pair sit on one XCD (blockIdx % 8).
// build + run:  hipcc -O3 --offload-arch=gfx950 scripts/ubench_atomics.hip -o /tmp/ubench_atomics && /tmp/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
  float* dq;        // [BH, L, D]
  int L, D, KB, nkb, steps;
  int per_step;     // floats a block adds per query step
  int stagger;      // key block kb starts at query step kb * stagger
  int spin;         // dependent FMAs between steps (stand-in for the tile's compute)
  int same_xcd;
};

template <int KIND>   // 0: atomic add (no return)  1: plain store (yard-stick)  2: plain load+add+store (non-atomic RMW, wrong but a yard-stick)
__global__ __launch_bounds__(512) void atom_kernel(Args p) {
  const int bid = blockIdx.x;
  int bh, kb;
  if (p.same_xcd) {
    const int xcd = bid & 7, slot = bid >> 3;
    bh = (slot / p.nkb) * 8 + xcd;
    kb = slot % p.nkb;
  } else {
    bh = bid / p.nkb;
    kb = bid % p.nkb;
  }
  float* base = p.dq + (size_t)bh * p.L * p.D;
  const int tid = threadIdx.x;
  float v = 1.0f + tid * 1e-6f;
  for (int s = 0; s < p.steps; ++s) {
    const int qs = (s + kb * p.stagger) % p.steps;
    float* dst = base + (size_t)qs * 32 * p.D;
    for (int i = tid; i < p.per_step; i += 512) {
      const int j = i % (32 * p.D);   // "pair" mode wraps: several waves add to the same [32, D] tile
      if (KIND == 0) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(dst + j), v);
      if (KIND == 1) dst[j] = v;
      if (KIND == 2) dst[j] += v;
    }
    for (int k = 0; k < p.spin; ++k) v = __builtin_fmaf(v, 0.9999f, 1e-4f);
  }
  if (v == 123.f) base[0] = v;
}

int main() {
  const int B = 8, H = 16, L = 1280, D = 128;
  float* dq;
  CK(hipMalloc(&dq, (size_t)B * H * L * D * 4));
  CK(hipMemset(dq, 0, (size_t)B * H * L * D * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Case { const char* name; int KB; int per_step_mult; int stagger; int spin; int same_xcd; };
  std::vector<Case> cases = {
      {"KB=128 block-reduced, lockstep", 128, 1, 0, 0, 1},
      {"KB=128 block-reduced, staggered", 128, 1, 4, 0, 1},
      {"KB=128 per-pair (4x), lockstep", 128, 4, 0, 0, 1},
      {"KB=128 per-pair (4x), staggered", 128, 4, 4, 0, 1},
      {"KB=256 block-reduced, staggered", 256, 1, 8, 0, 1},
      {"KB=128 block-reduced, staggered, any XCD", 128, 1, 4, 0, 0},
      {"KB=128 block-reduced, staggered, spin 2000", 128, 1, 4, 2000, 1},
      {"KB=128 per-pair (4x), staggered, spin 2000", 128, 4, 4, 2000, 1},
      {"KB=128 no atomics, spin 2000 only", 128, 0, 4, 2000, 1},
  };
  for (const Case& c : cases) {
    for (int kind = 0; kind < 3; ++kind) {
      Args a{dq, L, D, c.KB, L / c.KB, L / 32, 32 * D * c.per_step_mult, c.stagger, c.spin, c.same_xcd};
      const int blocks = B * H * a.nkb;
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(atom_kernel<0>, dim3(blocks), dim3(512), 0, 0, a);
        if (kind == 1) hipLaunchKernelGGL(atom_kernel<1>, dim3(blocks), dim3(512), 0, 0, a);
        if (kind == 2) hipLaunchKernelGGL(atom_kernel<2>, dim3(blocks), dim3(512), 0, 0, a);
      };
      for (int w = 0; w < 3; ++w) launch();
      CK(hipEventRecord(e0));
      const int reps = 10;
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1000.0 / reps;
      const double bytes = (double)blocks * a.steps * a.per_step * 4.0;
      printf("%-48s %-8s %8.1f us  %7.1f MB  %6.2f TB/s\n", c.name, kind == 0 ? "atomic" : kind == 1 ? "store" : "rmw", us, bytes / 1e6, bytes / us / 1e6);
    }
  }
  // correctness spot check of the atomic: every element of pair 0 got nkb adds per launch
  return 0;
}
