// Phase timeline of conv_tile32_kernel on gfx950 (kernel work; not part of the product).  Builds conv_tile.hip with
// -DADP_TILE_TRACE and prints, per stage of the staged waves, when each phase boundary is reached, plus the hardware placement
// of a workgroup's waves (which waves share a SIMD).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DADP_TILE_TRACE -I include -I audio_diffusion_pytorch_amd/csrc \
//          -o tools/probe/tile_probe tools/probe/tile_probe.hip
//   run:   ADP_TILE_CFG=1 tools/probe/tile_probe [tr] [pro] [res] [gn]
#include "../../audio_diffusion_pytorch_amd/csrc/conv_tile.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
void adp_rt_note_launch(const char*, const char*, void*) {}
void adp_rt_launch_done(void*) {}

int main(int argc, char** argv) {
  const int tr = argc > 1 ? atoi(argv[1]) : 1, pro = argc > 2 ? atoi(argv[2]) : 0, res = argc > 3 ? atoi(argv[3]) : 0,
            gn = argc > 4 ? atoi(argv[4]) : 0;
  const int64_t B = 4, C = 32, L = 65536, n = B * C * L;
  float *x, *out, *r, *w, *bias, *stats, *gnp, *gb;
  long long* ws;
  hipMalloc(&x, n * 4); hipMalloc(&out, n * 4); hipMalloc(&r, n * 4); hipMalloc(&w, 32 * 32 * 3 * 4); hipMalloc(&bias, 128);
  hipMalloc(&stats, B * 8 * 2 * 4); hipMalloc(&gnp, 1 << 20); hipMalloc(&gb, 256); hipMalloc(&ws, 4096 * 8 * 8);
  hipMemset(x, 0, n * 4); hipMemset(r, 0, n * 4); hipMemset(w, 0, 32 * 32 * 3 * 4); hipMemset(bias, 0, 128);
  std::vector<float> st(B * 8 * 2, 1.0f);
  hipMemcpy(stats, st.data(), st.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(gb, st.data(), 32 * 4, hipMemcpyHostToDevice);
  adp_conv_desc d{};
  d.x = x; d.w = w; d.bias = tr ? nullptr : bias; d.out = out; d.res = res ? r : nullptr;
  d.pro_stats = stats; d.pro_gamma = gb; d.pro_beta = gb; d.gn_part = gn ? gnp : nullptr; d.ws = (float*)ws;
  d.B = B; d.R = C; d.R1 = C; d.Lin = L; d.M = C; d.N = L; d.KT = 3; d.stride = 1; d.dil = 1; d.pad = 1; d.up = 1;
  d.transposed = tr; d.prologue = pro; d.groups = 8; d.store = 0; d.sp = 1;
  for (int rep = 0; rep < 3; ++rep) {
    adp_conv_tile(d, nullptr);
    hipDeviceSynchronize();
  }
  std::vector<long long> h(4096 * 8);
  hipMemcpy(h.data(), ws, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0], t1 = 0;
  for (int i = 0; i < 4096; ++i) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 5]); }
  printf("tr %d pro %d res %d gn %d cfg %s: kernel span %.2f us\n", tr, pro, res, gn, getenv("ADP_TILE_CFG") ? getenv("ADP_TILE_CFG") : "1", (t1 - t0) * 0.01);
  const char* names[6] = {"start", "barrier", "released", "tile in LDS", "MFMAs done", "stored"};
  for (int s = 0; s < 4; ++s) {
    printf("stage %d:", s);
    for (int k = 0; k < 6; ++k) {
      double sum = 0, mn = 1e9, mx = 0;
      int cnt = 0;
      for (int i = 0; i < 4096; ++i)
        if (((i & 15) >> 2) == s) { double v = (h[i * 8 + k] - t0) * 0.01; sum += v; mn = std::min(mn, v); mx = std::max(mx, v); ++cnt; }
      printf("  %s %.1f [%.1f, %.1f]", names[k], sum / cnt, mn, mx);
    }
    printf("\n");
  }
  for (int wg = 0; wg < 2; ++wg) {
    printf("wg %d placement (wave: se/cu/simd):", wg);
    for (int wv = 0; wv < 16; ++wv) {
      const unsigned id = (unsigned)h[(wg * 16 + wv) * 8 + 7];
      printf(" %d:%u/%u/%u", wv, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3);
    }
    printf("\n");
  }
  return 0;
}
