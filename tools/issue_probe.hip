// issue_probe.hip -- what ONE wave alone on a SIMD pays per instruction, by instruction mix.
//
// The step kernels run one wave per SIMD at the BASELINE batch sizes, so a launch lasts as long
// as its longest wave's instruction stream; the per-lane predicates of interact() compile to
// v_cmp -> SGPR-pair mask -> s_and/s_or -> v_cndmask chains.  This probe times, with s_memtime
// around 16 x 64 repetitions, a few streams of that shape against all-VALU formulations of the
// same logic (masks as 0 / -1 words in VGPRs, v_bfi selects, v_bitop3 three-input logic).
//     hipcc --offload-arch=gfx950 -O2 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REPS 64
#define LOOPS 16
#define STR2(x) #x
#define STR(x) STR2(x)

#define PROBE(NAME, BODY, NINSTR)                                                                  \
  __global__ void NAME(unsigned long long *out, int *sink, int seed) {                             \
    int a = threadIdx.x + seed, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13, f = 17;            \
    unsigned long long m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull, m2 = 0, m3 = 0;     \
    unsigned long long t0, t1;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                     \
    for (int it = 0; it < LOOPS; it++) {                                                           \
      asm volatile(".rept " STR(REPS) "\n" BODY "\n.endr"                                              \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+s"(m0), "+s"(m1),     \
                     "+s"(m2), "+s"(m3)::"vcc", "scc");                                            \
    }                                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                     \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                        \
    sink[threadIdx.x] = a + b + c + d + e + f + (int)m0 + (int)m1 + (int)m2 + (int)m3;             \
  }                                                                                                \
  static const int NAME##_n = NINSTR;

// operands: %0 a %1 b %2 c %3 d %4 e %5 f (VGPR); %6 m0 %7 m1 %8 m2 %9 m3 (SGPR pairs)
PROBE(p_valu_dep, "v_add_u32 %0, %0, %1", 1)
PROBE(p_valu_indep4, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4", 4)
PROBE(p_salu_dep, "s_lshl_b64 %6, %6, 1", 1)
PROBE(p_salu64_dep, "s_and_b64 %6, %6, %7\n s_or_b64 %6, %6, %8", 2)
PROBE(p_v_s_alternate_indep, "v_add_u32 %0, %0, %4\n s_and_b64 %8, %6, %7\n v_add_u32 %1, %1, %4\n s_or_b64 %9, %6, %7", 4)
// the compiler's shape: compare -> mask logic on the scalar unit -> select, each feeding the next
PROBE(p_cmp_sand_cnd_chain,
      "v_cmp_eq_u32_e64 %8, %0, %1\n s_and_b64 %8, %8, %6\n v_cndmask_b32_e64 %0, %2, %3, %8", 3)
// two such chains interleaved (what ILP buys)
PROBE(p_cmp_sand_cnd_x2,
      "v_cmp_eq_u32_e64 %8, %0, %1\n v_cmp_eq_u32_e64 %9, %4, %5\n s_and_b64 %8, %8, %6\n s_and_b64 %9, %9, %7\n"
      " v_cndmask_b32_e64 %0, %2, %3, %8\n v_cndmask_b32_e64 %4, %2, %3, %9", 6)
// compare -> select directly (no scalar logic in between)
PROBE(p_cmp_cnd_chain, "v_cmp_eq_u32_e64 %8, %0, %1\n v_cndmask_b32_e64 %0, %2, %3, %8", 2)
PROBE(p_cmp_vcc_cnd_chain, "v_cmp_eq_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %2, %3, vcc", 2)
// the same decision as VALU words: eq-mask = ((a ^ b) - 1) >> 31 (small non-negative operands), and, bfi select
PROBE(p_mask_valu_chain,
      "v_xad_u32 %4, %0, %1, -1\n v_ashrrev_i32 %4, 31, %4\n v_and_b32 %4, %4, %5\n v_bfi_b32 %0, %4, %2, %3", 4)
PROBE(p_mask_valu_x2,
      "v_xad_u32 %4, %0, %1, -1\n v_xad_u32 %5, %2, %3, -1\n v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n"
      " v_bfi_b32 %0, %4, %2, %3\n v_bfi_b32 %2, %5, %0, %1", 6)
// three-input logic in one instruction
PROBE(p_bitop3_dep, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4", 1)
PROBE(p_bfi_dep, "v_bfi_b32 %0, %1, %0, %2", 1)
// a compare whose SGPR result is consumed by SALU only, then a scalar branch-free use
PROBE(p_cmp_then_salu, "v_cmp_eq_u32_e64 %8, %0, %1\n s_and_b64 %9, %8, %6", 2)
PROBE(p_readlane_chain, "v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0", 2)
PROBE(p_nop, "s_nop 0", 1)

// vector-memory ISSUE cost: back-to-back row stores / loads of one wave (64 lanes x 4 B, write-through as the
// step kernels store), no wait in between: what ~30 observation row stores cost an observation wave
#define MEMPROBE(NAME, BODY, NINSTR)                                                               \
  __global__ void NAME(unsigned long long *out, int *sink, int seed) {                             \
    int v = threadIdx.x + seed, off = threadIdx.x * 4;                                             \
    typedef int v4i __attribute__((ext_vector_type(4)));                                           \
    v4i q = {v, v, v, v};                                                                          \
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(sink + 1024, 0, 1 << 20, 0x00020000); \
    unsigned long long t0, t1;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                     \
    asm volatile("s_mov_b32 s20, 0\n.rept 32\n" BODY "\n.endr" : "+v"(v), "+v"(q) : "v"(off), "s"(r) : "memory", "s20");   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    if (threadIdx.x == 0) out[0] = (t1 - t0) * (REPS * LOOPS) / 32;                                \
    sink[threadIdx.x] = v + q.x;                                                                   \
  }                                                                                                \
  static const int NAME##_n = NINSTR;
MEMPROBE(m_store_dword_sc1, "buffer_store_dword %0, %2, %3, 0 offen sc1", 1)
MEMPROBE(m_store_dword, "buffer_store_dword %0, %2, %3, 0 offen", 1)
MEMPROBE(m_store_dwordx4_sc1, "buffer_store_dwordx4 %1, %2, %3, 0 offen sc1", 1)
MEMPROBE(m_store_byte_sc1, "buffer_store_byte %0, %2, %3, 0 offen sc1", 1)
// the row stores of the step kernels: the row offset in a FRESHLY written scalar (soffset), as the kernels
// form it (s_mul_i32 per row), against the same offset added into the lane's VGPR offset
MEMPROBE(m_store_soffset_fresh, "s_add_u32 s20, s20, 64\n buffer_store_dword %0, %2, %3, s20 offen sc1", 2)
MEMPROBE(m_store_soffset_fresh_valu2, "s_add_u32 s20, s20, 64\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n buffer_store_dword %0, %2, %3, s20 offen sc1", 4)
MEMPROBE(m_store_soffset_early, "buffer_store_dword %0, %2, %3, s20 offen sc1\n s_add_u32 s20, s20, 64", 2)
MEMPROBE(m_store_voffset_fresh, "v_add_u32 %0, 64, %0\n buffer_store_dword %0, %0, %3, 0 offen sc1", 2)
MEMPROBE(m_store_dword_sc1_valu3, "buffer_store_dword %0, %2, %3, 0 offen sc1\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0", 4)
MEMPROBE(m_store_dword_sc1_valu7, "buffer_store_dword %0, %2, %3, 0 offen sc1\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0", 8)

// the same row stores from FOUR waves of one workgroup at once (one per SIMD of a CU, as a split launch has
// them): do the four share one address / store pipe?
__global__ void __launch_bounds__(256) m4_store_rows(unsigned long long *out, int *sink, int seed) {
  int v = threadIdx.x + seed, off = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 65536;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(sink + 1024, 0, 1 << 20, 0x00020000);
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_mov_b32 s20, 0\n.rept 32\n s_add_u32 s20, s20, 1024\n buffer_store_dword %0, %1, %2, s20 offen sc1\n.endr" : "+v"(v) : "v"(off), "s"(r) : "memory", "s20");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0);
  sink[threadIdx.x] = v;
}


// latency of the scalar time reads the TIMELINE build uses (and of a scalar kernarg-style load, for
// scale): s_memtime; <op>; s_waitcnt lgkmcnt(0); s_memtime -- minus the same with no <op>
__global__ void smem_latency(unsigned long long *out, const int *src) {
  unsigned long long t0, t1, t2, t3, t4, x;
  int y;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(y) : "s"(src) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(y) : "s"(src) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4)::"memory");
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;   // one s_memtime round trip
    out[1] = t2 - t1;   // s_memrealtime + s_memtime
    out[2] = t3 - t2;   // s_load_dword (cold) + s_memtime
    out[3] = t4 - t3;   // s_load_dword (scalar-cache hit) + s_memtime
    out[4] = x + y;
  }
}

#define RUN(NAME)                                                                               \
  do {                                                                                          \
    double best = 1e30;                                                                         \
    for (int r = 0; r < 5; r++) {                                                               \
      hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, out, sink, r);                          \
      hipDeviceSynchronize();                                                                   \
      unsigned long long h;                                                                     \
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);                                             \
      const double per = (double)h / (REPS * LOOPS);                                            \
      if (per < best) best = per;                                                               \
    }                                                                                           \
    printf("%-28s %2d instructions per repetition: %6.2f cycles per repetition = %5.2f per instruction\n", #NAME, \
           NAME##_n, best, best / NAME##_n);                                                    \
  } while (0)

int main() {
  unsigned long long *out;
  int *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 2 << 20);
  RUN(p_nop);
  RUN(p_valu_dep);
  RUN(p_valu_indep4);
  RUN(p_salu_dep);
  RUN(p_salu64_dep);
  RUN(p_v_s_alternate_indep);
  RUN(p_cmp_sand_cnd_chain);
  RUN(p_cmp_sand_cnd_x2);
  RUN(p_cmp_cnd_chain);
  RUN(p_cmp_vcc_cnd_chain);
  RUN(p_mask_valu_chain);
  RUN(p_mask_valu_x2);
  RUN(p_bitop3_dep);
  RUN(p_bfi_dep);
  RUN(p_cmp_then_salu);
  RUN(p_readlane_chain);
  RUN(m_store_dword_sc1);
  RUN(m_store_dword);
  RUN(m_store_dwordx4_sc1);
  RUN(m_store_byte_sc1);
  RUN(m_store_soffset_fresh);
  RUN(m_store_soffset_fresh_valu2);
  RUN(m_store_soffset_early);
  RUN(m_store_voffset_fresh);
  RUN(m_store_dword_sc1_valu3);
  RUN(m_store_dword_sc1_valu7);
  for (int waves = 1; waves <= 4; waves++) {
    double best = 1e30;
    for (int r = 0; r < 5; r++) {
      hipLaunchKernelGGL(m4_store_rows, dim3(1), dim3(64 * waves), 0, 0, out, sink, r);
      hipDeviceSynchronize();
      unsigned long long h[4];
      hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
      unsigned long long mx = 0;
      for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
      if ((double)mx < best) best = (double)mx;
    }
    printf("32 row stores (+ s_add each) per wave, %d wave(s) of one workgroup at once: slowest wave %.0f cycles = %.1f per store\n",
           waves, best, best / 32);
  }
  {
    unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    for (int r = 0; r < 5; r++) {
      hipLaunchKernelGGL(smem_latency, dim3(1), dim3(64), 0, 0, out, sink);
      hipDeviceSynchronize();
      unsigned long long h[5];
      hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
      for (int k = 0; k < 4; k++) best[k] = h[k] < best[k] ? h[k] : best[k];
    }
    printf("scalar time reads, issue -> result (shader cycles): s_memtime %llu, s_memrealtime %llu, s_load_dword cold %llu, hit %llu\n",
           best[0], best[1] - best[0], best[2] - best[0], best[3] - best[0]);
  }
  return 0;
}
