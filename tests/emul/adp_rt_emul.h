// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// A tiny host-side SIMT emulator so the *same* kernel sources under
// audio_diffusion_pytorch_amd/csrc/ can be compiled with g++ (-DADP_EMULATE) and their
// index / tiling / reduction logic exercised in the GPU-less build container.
// Every HIP thread of a workgroup is a ucontext fiber; __syncthreads() and the wave64
// collectives (__shfl*, MFMA) are rendezvous points.  Workgroups run one after another.
// The MFMA emulation follows the gfx950 fragment layouts documented in
// /opt/skills/guides/cdna_hip_programming.md section 3 (f32-input 32x32x2 and 16x16x4 forms)
// and accumulates with a k-ordered fmaf chain, which is what the hardware does.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct adp_uint3 {
  unsigned x, y, z;
};
struct float4 {
  float x, y, z, w;
};
struct float2 {
  float x, y;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
struct f32x16 {
  float v[16];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
struct f32x2 {
  float v[2];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
// elementwise arithmetic of the clang vector type (the product build compiles these to packed-f32 VALU ops)
static inline f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{{a[0] + b[0], a[1] + b[1]}}; }
static inline f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{{a[0] - b[0], a[1] - b[1]}}; }
static inline f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{{a[0] * b[0], a[1] * b[1]}}; }
static inline f32x2 operator*(f32x2 a, float b) { return f32x2{{a[0] * b, a[1] * b}}; }
static inline f32x2 operator+(f32x2 a, float b) { return f32x2{{a[0] + b, a[1] + b}}; }
static inline f32x2 operator-(f32x2 a, float b) { return f32x2{{a[0] - b, a[1] - b}}; }
struct f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
static inline f32x4 operator+(f32x4 a, f32x4 b) { return f32x4{{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
static inline f32x4 operator*(f32x4 a, float b) { return f32x4{{a[0] * b, a[1] * b, a[2] * b, a[3] * b}}; }
static inline f32x4 operator*(f32x4 a, f32x4 b) { return f32x4{{a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]}}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace adp_emul {

enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  int state = READY;
  unsigned coll = 0;  // per-lane count of wave collectives (double-buffer parity)
};

struct State {
  adp_uint3 tid{0, 0, 0}, bid{0, 0, 0};
  dim3 bdim, gdim;
  int cur = 0;  // linear thread index inside the block
  ucontext_t sched;
  std::vector<Fiber> fibers;
  std::vector<uint32_t> slotA, slotB;  // [wave][2][64]
  std::function<void()> body;
  size_t stack_size = 256 * 1024;
};

inline State& S() {
  static State s;
  return s;
}

inline void trampoline() {
  State& s = S();
  s.body();
  s.fibers[s.cur].state = DONE;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void yield_as(int st) {
  State& s = S();
  int me = s.cur;
  s.fibers[me].state = st;
  swapcontext(&s.fibers[me].ctx, &s.sched);
}

inline void sync_block() { yield_as(WAIT_BLOCK); }
inline void sync_wave() { yield_as(WAIT_WAVE); }

inline void set_tid(State& s, int t) {
  s.cur = t;
  s.tid.x = t % s.bdim.x;
  s.tid.y = (t / s.bdim.x) % s.bdim.y;
  s.tid.z = t / (s.bdim.x * s.bdim.y);
}

inline void run_block(State& s) {
  const int n = s.bdim.x * s.bdim.y * s.bdim.z;
  const int nwaves = (n + 63) / 64;
  if ((int)s.fibers.size() < n) s.fibers.resize(n);
  s.slotA.assign((size_t)nwaves * 2 * 64, 0);
  s.slotB.assign((size_t)nwaves * 2 * 64, 0);
  for (int t = 0; t < n; ++t) {
    Fiber& f = s.fibers[t];
    if (!f.stack) f.stack = (char*)malloc(s.stack_size);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = s.stack_size;
    f.ctx.uc_link = nullptr;
    f.state = READY;
    f.coll = 0;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  int ndone = 0;
  while (ndone < n) {
    bool progressed = false;
    for (int t = 0; t < n; ++t) {
      if (s.fibers[t].state != READY) continue;
      set_tid(s, t);
      swapcontext(&s.sched, &s.fibers[t].ctx);
      progressed = true;
      if (s.fibers[t].state == DONE) ++ndone;
    }
    // release wave rendezvous
    for (int w = 0; w < nwaves; ++w) {
      int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
      bool all = true, any = false;
      for (int t = lo; t < hi; ++t) {
        int st = s.fibers[t].state;
        if (st == WAIT_WAVE) any = true;
        else if (st != DONE) all = false;
      }
      if (all && any) {
        for (int t = lo; t < hi; ++t)
          if (s.fibers[t].state == WAIT_WAVE) s.fibers[t].state = READY;
        progressed = true;
      }
    }
    // release block barrier
    {
      bool all = true, any = false;
      for (int t = 0; t < n; ++t) {
        int st = s.fibers[t].state;
        if (st == WAIT_BLOCK) any = true;
        else if (st != DONE) all = false;
      }
      if (all && any) {
        for (int t = 0; t < n; ++t)
          if (s.fibers[t].state == WAIT_BLOCK) s.fibers[t].state = READY;
        progressed = true;
      }
    }
    if (!progressed) {
      fprintf(stderr, "adp_emul: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n", s.bid.x, s.bid.y,
              s.bid.z);
      abort();
    }
  }
}

inline void launch(dim3 grid, dim3 block, std::function<void()> body) {
  State& s = S();
  s.gdim = grid;
  s.bdim = block;
  s.body = std::move(body);
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.bid = adp_uint3{x, y, z};
        run_block(s);
      }
}

inline int lane_id() { return S().cur & 63; }
inline int wave_id() { return S().cur >> 6; }

template <typename T>
inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) == 4, "4-byte shuffles only");
  State& s = S();
  Fiber& f = s.fibers[s.cur];
  uint32_t* slot = &s.slotA[((size_t)wave_id() * 2 + (f.coll & 1)) * 64];
  uint32_t bits;
  memcpy(&bits, &v, 4);
  slot[lane_id()] = bits;
  ++f.coll;
  sync_wave();
  uint32_t r = slot[src & 63];
  T out;
  memcpy(&out, &r, 4);
  return out;
}

inline void mfma_exchange(float a, float b, const float*& A, const float*& B) {
  State& s = S();
  Fiber& f = s.fibers[s.cur];
  size_t base = ((size_t)wave_id() * 2 + (f.coll & 1)) * 64;
  memcpy(&s.slotA[base + lane_id()], &a, 4);
  memcpy(&s.slotB[base + lane_id()], &b, 4);
  ++f.coll;
  sync_wave();
  A = reinterpret_cast<const float*>(&s.slotA[base]);
  B = reinterpret_cast<const float*>(&s.slotB[base]);
}

}  // namespace adp_emul

#define threadIdx (adp_emul::S().tid)
#define blockIdx (adp_emul::S().bid)
#define blockDim (adp_emul::S().bdim)
#define gridDim (adp_emul::S().gdim)

static inline void __syncthreads() { adp_emul::sync_block(); }

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return adp_emul::shfl_idx(v, adp_emul::lane_id() ^ mask);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int l = adp_emul::lane_id();
  int src = ((l % width) + (int)delta < width) ? l + (int)delta : l;
  return adp_emul::shfl_idx(v, src);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  int l = adp_emul::lane_id();
  return adp_emul::shfl_idx(v, (l / width) * width + (src % width));
}

// wave vote (all 64 lanes active)
static inline int __all(int pred) {
  int v = pred != 0;
  for (int o = 32; o >= 1; o >>= 1) v &= __shfl_xor(v, o);
  return v;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
static inline f32x16 adp_mfma32(float a, float b, f32x16 c) {
  const float *A, *B;
  adp_emul::mfma_exchange(a, b, A, B);
  int l = adp_emul::lane_id();
  int col = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(A[row + 32 * k], B[col + 32 * k], acc);
    d[r] = acc;
  }
  return d;
}
// v_mfma_f32_32x32x16_bf16: lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], j = 0..7 (bf16); fp32 accumulate
struct bf16x8 {
  uint16_t v[8];
};
static inline float adp_emul_bf16(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline f32x16 adp_mfma32_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  int l = adp_emul::lane_id();
  int col = l & 31;
  f32x16 d = c;
  for (int w = 0; w < 4; ++w) {  // one 32-bit exchange per bf16 pair
    uint32_t aw, bw;
    memcpy(&aw, &a.v[2 * w], 4);
    memcpy(&bw, &b.v[2 * w], 4);
    float af, bf;
    memcpy(&af, &aw, 4);
    memcpy(&bf, &bw, 4);
    const float *A, *B;
    adp_emul::mfma_exchange(af, bf, A, B);
    for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float acc = d[r];
      for (int kh = 0; kh < 2; ++kh) {
        uint32_t ua, ub;
        memcpy(&ua, &A[row + 32 * kh], 4);
        memcpy(&ub, &B[col + 32 * kh], 4);
        acc = fmaf(adp_emul_bf16((uint16_t)(ua & 0xffff)), adp_emul_bf16((uint16_t)(ub & 0xffff)), acc);
        acc = fmaf(adp_emul_bf16((uint16_t)(ua >> 16)), adp_emul_bf16((uint16_t)(ub >> 16)), acc);
      }
      d[r] = acc;
    }
  }
  return d;
}
struct uint2 {
  uint32_t x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint32_t __float_as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float __uint_as_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; D: col=l&15, row=(l>>4)*4+r
static inline f32x4 adp_mfma16(float a, float b, f32x4 c) {
  const float *A, *B;
  adp_emul::mfma_exchange(a, b, A, B);
  int l = adp_emul::lane_id();
  int col = l & 15;
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(A[row + 16 * k], B[col + 16 * k], acc);
    d[r] = acc;
  }
  return d;
}

static inline float atomicAdd(float* p, float v) {
  float o = *p;
  *p = o + v;
  return o;
}
#define __expf(x) expf(x)
static inline float adp_rcp(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }

static inline void adp_barrier_consume() { adp_emul::sync_block(); }
static inline void adp_barrier_lds() { adp_emul::sync_block(); }
static inline void adp_sched_fence() {}
template <class T> static inline T adp_nt_load(const T* p) { return *p; }
template <class T> static inline void adp_nt_store(T v, T* p) { *p = v; }
static inline void adp_wave_sync() { adp_emul::sync_wave(); }
static inline void adp_keep(float, float) {}
static inline void adp_pin(f32x2&) {}
static inline void adp_setprio(int) {}
static inline float adp_exp2(float x) { return exp2f(x); }
static inline float adp_lane_prev(float edge, float v) {
  const int l = adp_emul::lane_id();
  const float o = adp_emul::shfl_idx(v, l > 0 ? l - 1 : 0);
  return l > 0 ? o : edge;
}
static inline float adp_lane_next(float edge, float v) {
  const int l = adp_emul::lane_id();
  const float o = adp_emul::shfl_idx(v, l < 63 ? l + 1 : 63);
  return l < 63 ? o : edge;
}
static inline float adp_row_prev(float v) {
  const int l = adp_emul::lane_id();
  const float o = adp_emul::shfl_idx(v, (l & 15) ? l - 1 : l);
  return (l & 15) ? o : 0.0f;
}
static inline float adp_row_next(float v) {
  const int l = adp_emul::lane_id();
  const float o = adp_emul::shfl_idx(v, (l & 15) != 15 ? l + 1 : l);
  return (l & 15) != 15 ? o : 0.0f;
}
static inline float adp_oct_sum(float v) {
  for (int o = 1; o < 8; o <<= 1) v += adp_emul::shfl_idx(v, adp_emul::lane_id() ^ o);
  return v;
}
static inline int adp_uniform(int v) { return v; }
static inline float adp_read_lane(float v, int src) { return adp_emul::shfl_idx(v, src); }
static inline float adp_row16_sum(float v) {
  for (int o = 1; o < 16; o <<= 1) v += adp_emul::shfl_idx(v, adp_emul::lane_id() ^ o);
  return v;
}
static inline float adp_half_sum(float v) {  // (valid in every lane here; the hardware form only in lanes 16-31 / 48-63)
  for (int o = 1; o < 32; o <<= 1) v += adp_emul::shfl_idx(v, adp_emul::lane_id() ^ o);
  return v;
}
static inline long long adp_clock() { return 0; }
static inline void adp_wait_until(long long) {}  // (the emulator runs a workgroup's waves one after another)

#define ADP_LAUNCH(kern, grid, block, stream, ...) \
  do {                                             \
    (void)(stream);                                \
    adp_rt_note_launch(#kern, __PRETTY_FUNCTION__, nullptr); \
    adp_emul::launch(grid, block, [=]() { kern(__VA_ARGS__); }); \
  } while (0)
#define ADP_LAUNCH_OK() 0
