// coissue_probe.hip -- how much other work a SIMD can issue under a stream of v_mfma_f32_16x16x4_f32
// (32 pipe cycles each) before the matrix pipe starts to starve.  Per loop iteration: 8 independent MFMAs,
// NV VALU ops (v_fma_f32) and NL LDS reads (ds_read_b64) spread evenly between them; FEED = the VALU results
// are the MFMAs' A operands (the shape of an on-the-fly operand transform).  1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, bool FEED, int KIND = 0>
__global__ __launch_bounds__(512, 1) void loop(float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float fb = 0.5f, c1 = 0.999f, c2 = 0.001f;
  float2 l2[4] = {};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) pk[i] = f32x2{threadIdx.x * 0.001f, i * 1.0f};
  int iv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) iv[i] = threadIdx.x + i;
  int sv = 0;
  const unsigned lp = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + (threadIdx.x & 63) * 2);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(FEED ? v[i] : c1, fb, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NV / 8; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 1 + k) & 7]) : "v"(c1), "v"(c2));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(i + 1 + k) & 3]) : "v"(pk[4]));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[(i + 1 + k) & 3]) : "v"(pk[4]));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[(i + 1 + k) & 7]) : "v"(iv[8]));
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[(i + 1 + k) & 7]) : "v"(iv[8]));
        if (KIND == 5) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sv));
        if (KIND == 6) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(lp), "v"(c1), "n"(8192 + 256 * ((i + k) & 3)) : "memory");
        if (KIND == 7) asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pk[(i + 1 + k) & 3]) : "v"(pk[4]));
        if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[(i + 1 + k) & 7]) : "v"(iv[8]) : );
      }
#pragma unroll
      for (int k = 0; k < (NL + 7 - i) / 8; ++k)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l2[(i + k) & 3]) : "v"(lp), "n"(512 * ((i + k) & 3)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += l2[i].x + l2[i].y + pk[i].x + pk[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += iv[i];
  s += sv;
  if (s == 123.456f) out[0] = s;
}

// vector-memory instructions under the MFMA stream: 32 MFMAs and NB instructions of KIND per iteration
// (0 buffer_load_dwordx4 into registers, 1 buffer_load_dword ... lds, 2 buffer_load_dwordx4 ... lds, 3 buffer_store_dwordx2),
// the loads of the PREVIOUS iteration awaited at the end of each (vmcnt(NB)); the source is 64 KB (L2 resident)
typedef int rsrc_words __attribute__((ext_vector_type(4)));
template <int KIND, int NB>
__global__ __launch_bounds__(512, 1) void loop_mem(float* out, const float* src, float* sink, int iters) {
  __shared__ float lds[8192];
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float fb = 0.5f, c1 = 0.999f;
  const unsigned long long a = (unsigned long long)(KIND == 3 ? sink : src);
  const rsrc_words rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)a),
                         (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu), 65536, 0x00020000};
  const unsigned voff = (threadIdx.x & 63) * 16;
  const unsigned lbase = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x >> 6) * 4096);
  f32x4 r[4] = {};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 st = {1.0f, 2.0f};
  for (int it = 0; it < iters; ++it) {
    const int soff = (it & 15) * 4096;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, fb, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if ((g * 8 + i) % (32 / NB) == 0) {
          if (KIND == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[g]) : "v"(voff), "s"(rs), "s"(soff) : "memory");
          if (KIND == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(lbase), "v"(voff), "s"(rs), "s"(soff) : "memory");
          if (KIND == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lbase), "v"(voff), "s"(rs), "s"(soff) : "memory");
          if (KIND == 3) asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" :: "v"(st), "v"(voff), "s"(rs), "s"(soff) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NB) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += r[i][0] + r[i][3];
  if (s == 123.456f) out[0] = s + lds[threadIdx.x];
}

// LDS-DMA (buffer_load_dword ... lds) in the shape conv3x3_winograd.hip issues it: 7 per 128 MFMAs, either as one
// burst or one per 16 MFMAs; lane addresses contiguous (256 B per instruction) or scattered like a raw 10 x 18
// patch (rows of 18 floats at a 448-byte pitch, two lanes in ten out of range); OFFS_LDS = the per-lane offsets
// come from LDS right before the burst (ds_read + lgkmcnt(0)), as the kernel does it
template <bool BURST, bool SCATTER, bool OFFS_LDS>
__global__ __launch_bounds__(512, 1) void loop_dma(float* out, const float* src, int iters) {
  __shared__ float lds[8192];
  __shared__ unsigned offs[8 * 7 * 64];
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float fb = 0.5f, c1 = 0.999f;
  const unsigned long long a = (unsigned long long)src;
  const rsrc_words rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)a),
                         (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu), 1 << 22, 0x00020000};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned vo[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int e = (wave + 8 * j) * 64 + lane;
    const int ch = e / 200, rem = e % 200, r = rem / 20, q = rem % 20;
    vo[j] = SCATTER ? (q < 18 ? (unsigned)((ch * 8960 + r * 112 + q) * 4) : 0x80000000u) : (unsigned)(e * 4);
    offs[(wave * 7 + j) * 64 + lane] = vo[j];
  }
  __syncthreads();
  const unsigned lbase = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x >> 6) * 256);
  for (int it = 0; it < iters; ++it) {
    const int soff = (it & 15) * 16 * 8960 * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, fb, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BURST ? g == 0 : g < 7) {
        if (OFFS_LDS && (BURST || g == 0)) {
#pragma unroll
          for (int j = 0; j < 7; ++j) vo[j] = offs[(wave * 7 + j) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < 7; ++j)
          if (BURST || j == g)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                         :: "s"(lbase + j * 2048), "v"(vo[j]), "s"(rs), "s"(soff) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s + lds[threadIdx.x];
}

template <bool BURST, bool SCATTER, bool OFFS_LDS>
void run_dma() {
  float *out, *src;
  hipMalloc(&out, 4); hipMalloc(&src, 1 << 23);
  hipMemset(src, 0, 1 << 23);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_dma<BURST, SCATTER, OFFS_LDS>), dim3(cus), dim3(512), 0, 0, out, src, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((loop_dma<BURST, SCATTER, OFFS_LDS>), dim3(cus), dim3(512), 0, 0, out, src, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_chunk = ms * 1e-3 * 2.4e9 / iters;      // SIMD cycles per 2 x 128 MFMAs
  printf("dma burst %d scatter %d offsets-from-LDS %d: %7.3f ms  %.0f cycles per chunk (8192 = MFMA only): +%.0f per wave\n",
         (int)BURST, (int)SCATTER, (int)OFFS_LDS, ms, per_chunk, (per_chunk - 8192 * 33.3 / 32) / 2);
  hipFree(out); hipFree(src);
}

template <int KIND, int NB>
void run_mem(int waves_per_simd) {
  float *out, *src, *sink;
  hipMalloc(&out, 4); hipMalloc(&src, 1 << 20); hipMalloc(&sink, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 5000;
  const dim3 grid(cus), block(256 * waves_per_simd);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_mem<KIND, NB>), grid, block, 0, 0, out, src, sink, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((loop_mem<KIND, NB>), grid, block, 0, 0, out, src, sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double need = (double)waves_per_simd * iters * 32 * 32;
  const double per = ms * 1e-3 * 2.4e9 / need * 32;
  printf("mem kind %d, %2d per 32 MFMAs, waves/SIMD %d: %7.3f ms  %.1f cycles per MFMA  (+%.1f cycles per memory instruction over 33.3)\n",
         KIND, NB, waves_per_simd, ms, per, (per - 33.3) * 32 / NB);
  hipFree(out); hipFree(src); hipFree(sink);
}

template <int NV, int NL, bool FEED, int KIND = 0>
void run(int waves_per_simd) {
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 20000;
  const dim3 grid(cus), block(256 * waves_per_simd);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<NV, NL, FEED, KIND>), grid, block, 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((loop<NV, NL, FEED, KIND>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // pipe cycles needed per SIMD: waves x iters x 8 MFMAs x 32
  const double need = (double)waves_per_simd * iters * 8 * 32;
  const double tf = 2.0 * 16 * 16 * 4 * 8.0 * iters * waves_per_simd * 4 * cus / ms / 1e9;
  printf("kind %d NV %2d NL %2d feed %d waves/SIMD %d: %7.3f ms  %6.1f TF/s = %.3f of 157.3   (%.1f cycles per MFMA at 2.4 GHz)\n",
         KIND, NV, NL, (int)FEED, waves_per_simd, ms, tf, tf / 157.3, ms * 1e-3 * 2.4e9 / need * 32);
  hipFree(out);
}

int main() {
  run_dma<true, false, false>(); run_dma<true, true, false>(); run_dma<true, true, true>();
  run_dma<false, false, false>(); run_dma<false, true, false>(); run_dma<false, true, true>();
  printf("mem kinds: 0 buffer_load_dwordx4, 1 buffer_load_dword lds, 2 buffer_load_dwordx4 lds, 3 buffer_store_dwordx2\n");
  run_mem<0, 2>(2); run_mem<0, 4>(2); run_mem<0, 8>(2);
  run_mem<1, 2>(2); run_mem<1, 4>(2); run_mem<1, 8>(2);
  run_mem<2, 2>(2); run_mem<2, 4>(2); run_mem<2, 8>(2);
  run_mem<3, 2>(2); run_mem<3, 4>(2); run_mem<3, 8>(2);
  printf("kinds: 0 v_fma_f32, 1 v_pk_add_f32, 2 v_pk_fma_f32, 3 v_add_u32, 4 v_mov_b32, 5 s_add_u32, 6 ds_write_b32, 7 v_pk_add_f32 with op_sel/neg, 8 v_cndmask_b32\n");
  for (int w = 2; w <= 2; ++w) {
    run<16, 0, false, 1>(w); run<32, 0, false, 1>(w);
    run<16, 0, false, 2>(w); run<32, 0, false, 2>(w);
    run<16, 0, false, 3>(w); run<32, 0, false, 3>(w);
    run<16, 0, false, 4>(w); run<32, 0, false, 4>(w);
    run<16, 0, false, 5>(w); run<32, 0, false, 5>(w);
    run<8, 0, false, 6>(w); run<16, 0, false, 6>(w);
    run<16, 0, false, 7>(w); run<32, 0, false, 7>(w);
    run<16, 0, false, 8>(w); run<32, 0, false, 8>(w);
  }
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, false>(w);
    run<8, 0, false>(w);
    run<16, 0, false>(w);
    run<32, 0, false>(w);
    run<48, 0, false>(w);
    run<16, 0, true>(w);
    run<32, 0, true>(w);
    run<0, 4, false>(w);
    run<0, 8, false>(w);
    run<16, 4, true>(w);
    run<16, 8, true>(w);
    run<32, 8, true>(w);
  }
  return 0;
}
