// v-objective elementwise math and small helpers (HBM-bound streaming kernels, gfx950).
//   adp_v_noise  : diffusion.py:88-92   x_noisy = a x + b n ; v_target = a n - b x      (2 reads, 2 writes)
//   adp_mse_*    : diffusion.py:95      F.mse_loss(v_pred, v_target) and its gradient
//   adp_v_step   : diffusion.py:185-187 one VSampler update, 2 reads 1 write
//   adp_time_fourier_* : a_unet NumberEmbedder under TimeConditioningPlugin (components.py:74-76)
#include "adp_rt.h"
#ifndef ADP_EMULATE
#include <mutex>
#endif
#include "adp.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;

__global__ __launch_bounds__(256) void v_noise_kernel(const float* x, const float* noise, const float* sigma,
                                                      int64_t per, float* x_noisy, float* v_target) {
  const int64_t b = blockIdx.y;
  // angle = sigma * pi / 2 evaluated left to right in fp32, as the reference does (diffusion.py:78)
  const float angle = (sigma[b] * PI_F) / 2.0f;
  const float a = cosf(angle), bt = sinf(angle);
  const int64_t base = b * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const float xv = x[base + i], nv = noise[base + i];
    x_noisy[base + i] = a * xv + bt * nv;
    v_target[base + i] = a * nv - bt * xv;
  }
}

constexpr int MSE_BLOCKS = 1024;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* p, const float* t, int64_t n, float* ws) {
  __shared__ float sh[4];
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = p[i] - t[i];
    s = fmaf(d, d, s);
  }
  s = adp_block_sum<4>(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

// one wave: lane l sums partials l, l + 64, ... in double (independent loads: a single thread walking all 1024 partials
// took 44 us), then the 64 lane sums are added in lane order by lane 0 -- fixed order, deterministic
__global__ __launch_bounds__(64) void mse_final_kernel(const float* ws, int nb, int64_t n, float* loss) {
  __shared__ double part[64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += (double)ws[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < 64; ++i) t += part[i];
  loss[0] = (float)(t / (double)n);
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* p, const float* t, const float* gloss, int64_t n,
                                                      float* dv) {
  const float sc = (gloss ? gloss[0] : 1.0f) * 2.0f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dv[i] = (p[i] - t[i]) * sc;
}

__global__ __launch_bounds__(256) void v_step_kernel(const float* x, const float* v, const float* ab4, int64_t n,
                                                     float* xo) {
  const float a0 = ab4[0], b0 = ab4[1], a1 = ab4[2], b1 = ab4[3];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], vv = v[i];
    const float x_pred = a0 * xv - b0 * vv;
    const float n_pred = b0 * xv + a0 * vv;
    xo[i] = a1 * x_pred + b1 * n_pred;
  }
}

// one VInpainter resample step (diffusion.py:339-350): rotate (x, v) from noise level i to level j, re-noise the
// source to level j with the caller's draw, keep the source where mask is set
__global__ __launch_bounds__(256) void v_inpaint_kernel(const float* x, const float* v, const float* src,
                                                        const float* noise, const uint8_t* mask, const float* ab4,
                                                        int64_t n, float* xo) {
  const float a0 = ab4[0], b0 = ab4[1], a1 = ab4[2], b1 = ab4[3];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], vv = v[i];
    const float x_pred = a0 * xv - b0 * vv;
    const float n_pred = b0 * xv + a0 * vv;
    const float xn = a1 * x_pred + b1 * n_pred;
    const float sn = a1 * src[i] + b1 * noise[i];
    xo[i] = mask[i] ? sn : xn;
  }
}

// classifier-free guidance mix of the two halves of a batched [2B, ...] evaluation:
// out = o_masked + (o - o_masked) * scale,  o = y[:half], o_masked = y[half:]
__global__ __launch_bounds__(256) void cfg_mix_kernel(const float* y, int64_t half, float scale, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (int64_t)gridDim.x * 256) {
    const float o = y[i], om = y[half + i];
    out[i] = om + (o - om) * scale;
  }
}

// out[b, :] = pick[b] ? a[b, :] : bsrc[b, :]   (per-row select; CFG's training-time embedding mask and its batch
// doubling write through it)
__global__ __launch_bounds__(256) void select_rows_kernel(const float* a, const float* bsrc, const uint8_t* pick,
                                                          int64_t rows, int64_t per, float* out) {
  const int64_t n = rows * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per;
    out[i] = pick[r] ? a[i] : bsrc[i];
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, int64_t n, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a[i] + b[i];
}

// out = a * x + b * y (y optional): gradient streams that meet with a constant factor (SkipCat's 2^-1/2 branch)
__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* x, float b, const float* y, int64_t n,
                                                    float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = y ? fmaf(a, x[i], b * y[i]) : a * x[i];
}

// dst[r * dst_stride + c] = src[r * src_stride + c]: channel concat / split of [B, C, L] tensors seen as B rows
__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, int64_t src_stride, float* dst,
                                                     int64_t dst_stride, int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * dst_stride + c] = src[r * src_stride + c];
  }
}

// space-to-depth of a [rows, L] tensor by factor f: out[(row * f + k), l] = x[row, l * f + k]  (a kernel = stride = f
// DownsampleItem with a factor the strided conv kernels do not cover becomes a 1x1 conv over the result)
__global__ __launch_bounds__(256) void unshuffle_kernel(const float* x, int64_t rows, int64_t Lo, int64_t f,
                                                        float* out) {
  const int64_t n = rows * Lo * f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / (Lo * f), p = i - row * (Lo * f);  // reads are coalesced
    const int64_t l = p / f, k = p - l * f;
    out[(row * f + k) * Lo + l] = x[i];
  }
}

// out[row, l] = sum_{k<f} x[row, l * f + k] (+ res[row, l]): gradient of a nearest upsample by any factor
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* x, int64_t rows, int64_t Lo, int64_t f,
                                                       const float* res, float* out) {
  const int64_t n = rows * Lo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float* p = x + i * f;
    float s = 0.0f;
    for (int64_t k = 0; k < f; ++k) s += p[k];
    out[i] = res ? s + res[i] : s;
  }
}

__device__ __forceinline__ float act_f(float x, int act) {
  return act == 1 ? adp_silu(x) : (act == 2 ? adp_gelu(x) : x);
}
__device__ __forceinline__ float dact_f(float x, int act) {
  return act == 1 ? adp_dsilu(x) : (act == 2 ? adp_dgelu(x) : 1.0f);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* x, int64_t n, int act, float* y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = act_f(x[i], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* x, const float* dy, int64_t n, int act,
                                                      int accumulate, float* dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float v = dy[i] * dact_f(x[i], act);
    dx[i] = accumulate ? dx[i] + v : v;
  }
}

// four[b, :] = [t, sin(f_0..f_{H-1}), cos(f_0..f_{H-1})],  f_h = ((t * w_h) * 2) * pi  (a_unet order of operations)
__global__ __launch_bounds__(256) void fourier_fwd_kernel(const float* t, const float* w, int64_t B, int64_t H,
                                                          float* four) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * H) return;
  const int64_t b = i / H, h = i % H;
  const float tv = t[b];
  const float f = ((tv * w[h]) * 2.0f) * PI_F;
  float* row = four + b * (2 * H + 1);
  if (h == 0) row[0] = tv;
  row[1 + h] = sinf(f);
  row[1 + H + h] = cosf(f);
}
__global__ __launch_bounds__(256) void fourier_bwd_kernel(const float* t, const float* w, const float* dfour,
                                                          int64_t B, int64_t H, int accumulate, float* dw) {
  const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  float s = 0.0f;
  for (int64_t b = 0; b < B; ++b) {
    const float tv = t[b];
    const float f = ((tv * w[h]) * 2.0f) * PI_F;
    const float* row = dfour + b * (2 * H + 1);
    s += (row[1 + h] * cosf(f) - row[1 + H + h] * sinf(f)) * (tv * 2.0f * PI_F);
  }
  dw[h] = accumulate ? dw[h] + s : s;
}

unsigned stream_grid(int64_t n) {
  int64_t g = adp_cdiv(n, 256 * 4);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int adp_version(void) { return 201; }

// ---- launch trace (profiling introspection; off by default, host-side only, per thread)
namespace {
thread_local bool g_trace_on = false;
thread_local char g_trace[4096];
thread_local size_t g_trace_len = 0;
#ifndef ADP_EMULATE
// The event list is process-wide (autograd runs the backward's launches on its own thread; the host code is
// serialised by the interpreter lock, so launches are appended in launch order).
constexpr int TRACE_EV_MAX = 8192;          // launches timed between two adp_launch_times calls
hipEvent_t g_ev[2 * TRACE_EV_MAX];
int g_ev_n = 0;                             // events recorded (two per launch)
std::mutex g_ev_mu;
#endif
}  // namespace

void adp_rt_note_launch(const char* kern, const char* site, void* stream) {
  if (!g_trace_on) return;
  const char* parts[4] = {g_trace_len ? "\n" : "", kern, "@", site};
  for (const char* p : parts)
    for (; *p && g_trace_len + 1 < sizeof(g_trace); ++p) g_trace[g_trace_len++] = *p;
  g_trace[g_trace_len] = 0;
#ifndef ADP_EMULATE
  std::lock_guard<std::mutex> lock(g_ev_mu);
  if (g_ev_n + 2 <= 2 * TRACE_EV_MAX) {
    hipEvent_t e;
    if (hipEventCreate(&e) == hipSuccess) {
      (void)hipEventRecord(e, (hipStream_t)stream);   // on the stream the kernel is launched on
      g_ev[g_ev_n++] = e;
    }
  }
#else
  (void)stream;
#endif
}

void adp_rt_launch_done(void* stream) {
#ifndef ADP_EMULATE
  if (!g_trace_on) return;
  std::lock_guard<std::mutex> lock(g_ev_mu);
  if ((g_ev_n & 1) == 0) return;
  hipEvent_t e;
  if (hipEventCreate(&e) == hipSuccess) {
    (void)hipEventRecord(e, (hipStream_t)stream);
    g_ev[g_ev_n++] = e;
  } else {  // keep the pairs aligned
    (void)hipEventDestroy(g_ev[--g_ev_n]);
  }
#else
  (void)stream;
#endif
}

extern "C" int64_t adp_launch_trace(int64_t enable, char* buf, int64_t cap) {
  int64_t n = 0;
  if (buf && cap > 0) {
    for (; n + 1 < cap && (size_t)n < g_trace_len; ++n) buf[n] = g_trace[n];
    buf[n] = 0;
  }
  g_trace_len = 0;
  g_trace[0] = 0;
  g_trace_on = enable != 0;
  return n;
}

extern "C" int64_t adp_launch_times(float* ms, int64_t cap) {
#ifndef ADP_EMULATE
  std::lock_guard<std::mutex> lock(g_ev_mu);
  const int pairs = g_ev_n / 2;
  int64_t n = 0;
  for (int i = 0; i < pairs; ++i) {
    float t = -1.0f;
    if (hipEventSynchronize(g_ev[2 * i + 1]) == hipSuccess) (void)hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]);
    if (ms && n < cap) ms[n++] = t;
    (void)hipEventDestroy(g_ev[2 * i]);
    (void)hipEventDestroy(g_ev[2 * i + 1]);
  }
  if (g_ev_n & 1) (void)hipEventDestroy(g_ev[g_ev_n - 1]);
  g_ev_n = 0;
  return n;
#else
  (void)ms;
  (void)cap;
  return 0;
#endif
}

extern "C" int adp_v_noise(const float* x, const float* noise, const float* sigma, int64_t B, int64_t per,
                           float* x_noisy, float* v_target, void* stream) {
  if (!x || !noise || !sigma || !x_noisy || !v_target) return ADP_ERR_NULL;
  if (B <= 0 || per <= 0 || B > 65535) return ADP_ERR_SHAPE;
  unsigned gx = stream_grid(per);
  ADP_LAUNCH(v_noise_kernel, dim3(gx, (unsigned)B), dim3(256), stream, x, noise, sigma, per, x_noisy, v_target);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_mse_ws_bytes(int64_t n) {
  (void)n;
  return MSE_BLOCKS * (int64_t)sizeof(float);
}

extern "C" int adp_mse_fwd(const float* v_pred, const float* v_target, int64_t n, float* loss, float* ws,
                           void* stream) {
  if (!v_pred || !v_target || !loss || !ws) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  int nb = (int)adp_cdiv(n, 256 * 8);
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  ADP_LAUNCH(mse_partial_kernel, dim3((unsigned)nb), dim3(256), stream, v_pred, v_target, n, ws);
  ADP_LAUNCH(mse_final_kernel, dim3(1), dim3(64), stream, (const float*)ws, nb, n, loss);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_mse_bwd(const float* v_pred, const float* v_target, const float* gloss, int64_t n, float* dv,
                           void* stream) {
  if (!v_pred || !v_target || !dv) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(mse_bwd_kernel, dim3(stream_grid(n)), dim3(256), stream, v_pred, v_target, gloss, n, dv);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_v_step(const float* x, const float* v, const float* ab4, int64_t n, float* x_out, void* stream) {
  if (!x || !v || !ab4 || !x_out) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(v_step_kernel, dim3(stream_grid(n)), dim3(256), stream, x, v, ab4, n, x_out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_v_inpaint_step(const float* x, const float* v, const float* source, const float* noise,
                                  const uint8_t* mask, const float* ab4, int64_t n, float* x_out, void* stream) {
  if (!x || !v || !source || !noise || !mask || !ab4 || !x_out) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(v_inpaint_kernel, dim3(stream_grid(n)), dim3(256), stream, x, v, source, noise, mask, ab4, n, x_out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_cfg_mix(const float* y, int64_t half, float scale, float* out, void* stream) {
  if (!y || !out) return ADP_ERR_NULL;
  if (half <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(cfg_mix_kernel, dim3(stream_grid(half)), dim3(256), stream, y, half, scale, out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_select_rows(const float* a, const float* b, const uint8_t* pick, int64_t rows, int64_t per,
                               float* out, void* stream) {
  if (!a || !b || !pick || !out) return ADP_ERR_NULL;
  if (rows <= 0 || per <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(select_rows_kernel, dim3(stream_grid(rows * per)), dim3(256), stream, a, b, pick, rows, per, out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_add(const float* a, const float* b, int64_t n, float* y, void* stream) {
  if (!a || !b || !y) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(add_kernel, dim3(stream_grid(n)), dim3(256), stream, a, b, n, y);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_axpby(float a, const float* x, float b, const float* y, int64_t n, float* out, void* stream) {
  if (!x || !out) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(axpby_kernel, dim3(stream_grid(n)), dim3(256), stream, a, x, b, y, n, out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_copy2d(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows,
                          int64_t cols, void* stream) {
  if (!src || !dst) return ADP_ERR_NULL;
  if (rows <= 0 || cols <= 0 || src_stride < cols || dst_stride < cols) return ADP_ERR_SHAPE;
  ADP_LAUNCH(copy2d_kernel, dim3(stream_grid(rows * cols)), dim3(256), stream, src, src_stride, dst, dst_stride, rows,
             cols);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_unshuffle(const float* x, int64_t rows, int64_t L, int64_t f, float* out, void* stream) {
  if (!x || !out) return ADP_ERR_NULL;
  if (rows <= 0 || L <= 0 || f < 1 || L % f != 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(unshuffle_kernel, dim3(stream_grid(rows * L)), dim3(256), stream, x, rows, L / f, f, out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_pool_sum(const float* x, int64_t rows, int64_t Lout, int64_t f, const float* res, float* out,
                            void* stream) {
  if (!x || !out) return ADP_ERR_NULL;
  if (rows <= 0 || Lout <= 0 || f < 1) return ADP_ERR_SHAPE;
  ADP_LAUNCH(pool_sum_kernel, dim3(stream_grid(rows * Lout)), dim3(256), stream, x, rows, Lout, f, res, out);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_act_fwd(const float* x, int64_t n, int64_t act, float* y, void* stream) {
  if (!x || !y) return ADP_ERR_NULL;
  if (n <= 0 || act < 0 || act > 2) return ADP_ERR_SHAPE;
  ADP_LAUNCH(act_fwd_kernel, dim3((unsigned)adp_cdiv(n, 256)), dim3(256), stream, x, n, (int)act, y);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_act_bwd(const float* x, const float* dy, int64_t n, int64_t act, int64_t accumulate, float* dx,
                           void* stream) {
  if (!x || !dy || !dx) return ADP_ERR_NULL;
  if (n <= 0 || act < 0 || act > 2) return ADP_ERR_SHAPE;
  ADP_LAUNCH(act_bwd_kernel, dim3((unsigned)adp_cdiv(n, 256)), dim3(256), stream, x, dy, n, (int)act,
             (int)accumulate, dx);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_time_fourier_fwd(const float* t, const float* w, int64_t B, int64_t H, float* four, void* stream) {
  if (!t || !w || !four) return ADP_ERR_NULL;
  if (B <= 0 || H <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(fourier_fwd_kernel, dim3((unsigned)adp_cdiv(B * H, 256)), dim3(256), stream, t, w, B, H, four);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_time_fourier_bwd(const float* t, const float* w, const float* dfour, int64_t B, int64_t H,
                                    int64_t accumulate, float* dw, void* stream) {
  if (!t || !w || !dfour || !dw) return ADP_ERR_NULL;
  if (B <= 0 || H <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(fourier_bwd_kernel, dim3((unsigned)adp_cdiv(H, 256)), dim3(256), stream, t, w, dfour, B, H,
             (int)accumulate, dw);
  return ADP_LAUNCH_OK();
}
