// Windowed-sinc polyphase resampler (utils.resample / downsample / upsample of
// /root/reference/audio_diffusion_pytorch/utils.py:82-117; the DiffusionUpsampler path, models.py:149-153, :163;
// SURVEY.md 8f-1).  The reference pads the waveform, runs ONE strided conv1d with `fo` output channels (one per
// output phase) and interleaves the phases with a rearrange "(b c) k l -> b c (l k)"; here the padding, the
// interleave and the final crop are index arithmetic inside one streaming kernel:
//
//   out[row, l*fo + k] = sum_{j < J} kern[k*J + j] * xpad[row, l*fi + j],   xpad[i] = x[i - width] or 0
//
// A workgroup owns 256 consecutive outputs of one row; the input span they touch ((256/fo)*fi + J samples) and the
// fo x J coefficient table are staged in LDS with coalesced loads, then every lane runs its J-tap dot product from
// LDS (lanes of one phase read consecutive addresses, a phase's coefficients are broadcast).
#include "adp_rt.h"
#include "adp.h"

namespace {

constexpr int RS_OUT = 256;      // outputs per workgroup
constexpr int RS_XCAP = 6144;    // floats of input span staged in LDS
constexpr int RS_KCAP = 4096;    // floats of coefficient table staged in LDS

__global__ __launch_bounds__(256) void resample_kernel(const float* x, const float* kern, int64_t length, int fi, int fo,
                                                       int J, int width, int64_t out_len, float* out) {
  __shared__ float xs[RS_XCAP];
  __shared__ float ks[RS_KCAP];
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.y;
  const int64_t o0 = (int64_t)blockIdx.x * RS_OUT;
  const int64_t o_last = (o0 + RS_OUT - 1 < out_len - 1) ? o0 + RS_OUT - 1 : out_len - 1;
  const int64_t l_lo = o0 / fo, l_hi = o_last / fo;
  const int64_t i0 = l_lo * fi - width;                       // first input index of the span (may be < 0)
  const int span = (int)((l_hi - l_lo) * fi + J);
  const float* xr = x + row * length;
  for (int i = tid; i < span; i += 256) {
    const int64_t g = i0 + i;
    xs[i] = (g >= 0 && g < length) ? xr[g] : 0.0f;              // the zero padding of F.pad
  }
  for (int i = tid; i < fo * J; i += 256) ks[i] = kern[i];
  __syncthreads();
  const int64_t o = o0 + tid;
  if (o >= out_len) return;
  const int64_t l = o / fo;
  const int k = (int)(o - l * fo);
  const float* xp = xs + (l - l_lo) * fi;
  const float* kp = ks + k * J;
  float acc = 0.0f;
  for (int j = 0; j < J; ++j) acc = fmaf(kp[j], xp[j], acc);
  out[row * out_len + o] = acc;
}

}  // namespace

extern "C" int adp_resample(const float* x, const float* kern, int64_t rows, int64_t length, int64_t fi, int64_t fo,
                            int64_t J, int64_t width, int64_t out_len, float* out, void* stream) {
  if (!x || !kern || !out) return ADP_ERR_NULL;
  if (rows <= 0 || length <= 0 || fi < 1 || fo < 1 || J < 1 || width < 0 || out_len <= 0) return ADP_ERR_SHAPE;
  if (rows > 65535 || length >= (int64_t)1 << 40) return ADP_ERR_SHAPE;
  // every output must be one the reference's conv produces: l <= (length + 2*width + fi - J) / fi
  if ((out_len - 1) / fo > (length + 2 * width + fi - J) / fi) return ADP_ERR_SHAPE;
  if (fo * J > RS_KCAP || ((RS_OUT - 1) / fo + 1) * fi + J > RS_XCAP) return ADP_ERR_UNSUPPORTED;
  dim3 grid((unsigned)adp_cdiv(out_len, RS_OUT), (unsigned)rows);
  ADP_LAUNCH(resample_kernel, grid, dim3(256), stream, x, kern, length, (int)fi, (int)fo, (int)J, (int)width, out_len, out);
  return ADP_LAUNCH_OK();
}
