// Host check of flvis_amd/csrc/eig_strip.hpp (the strip-mined corner-response phases): emulates the workgroup of k_eig_cand_strip tile
// by tile -- tile load with REFLECT_101, phase A for every item, phase B for every item -- over a whole image and compares every
// response that lies inside the image with a reference map, bit for bit; phase C (the 3x3 maxima as sort keys + the tile maximum)
// is compared with the one-pixel loop of k_eig_cand run on the same response tile.  Built and run by tests/test_eig_strip.py (g++, no GPU).
//
//   eig_strip_check <w> <h> <image.u8> <reference.f32>      exit 0: identical; 1: mismatch (count printed)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "eig_strip.hpp"

using namespace flvis::eigstrip;

static int reflect101c(int i, int n) {
  i = reflect101(i, n);
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
  std::vector<uint8_t> img((size_t)w * h);
  std::vector<float> ref((size_t)w * h);
  FILE* f = std::fopen(argv[3], "rb");
  if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) return 2;
  std::fclose(f);
  f = std::fopen(argv[4], "rb");
  if (!f || std::fread(ref.data(), 4, ref.size(), f) != ref.size()) return 2;
  std::fclose(f);
  std::vector<uint8_t> tile((size_t)IH * IW);
  std::vector<float> sfx((size_t)CH * CW), sfy((size_t)CH * CW), eig((size_t)OH * OW);
  long long checked = 0, bad = 0, keys_total = 0;
  for (int y0 = 0; y0 < h; y0 += TH)
    for (int x0 = 0; x0 < w; x0 += TW) {
      for (int r = 0; r < IH; r++)  // load_tile_u8<IH, IW, IW>(img, w, h, pitch, x0 - XOFF, y0 - HALO - 2, tile)
        for (int c = 0; c < IW; c++) tile[(size_t)r * IW + c] = img[(size_t)reflect101c(y0 - HALO - 2 + r, h) * w + reflect101c(x0 - XOFF + c, w)];
      std::memset(sfx.data(), 0xEE, sfx.size() * 4);  // every value must be written by the phases
      std::memset(sfy.data(), 0xEE, sfy.size() * 4);
      std::memset(eig.data(), 0xEE, eig.size() * 4);
      for (int item = 0; item < A_ITEMS; item++) sobel_strip(item, w, h, x0, y0, tile.data(), sfx.data(), sfy.data());
      for (int item = 0; item < B_ITEMS; item++) box_strip(item, sfx.data(), sfy.data(), eig.data());
      // phase C against the one-pixel loop of k_eig_cand on the same response tile: the same keys (as a set) and the same maximum
      {
        std::vector<unsigned long long> got, want;
        uint32_t gmax = 0, wmax = 0;
        for (int item = 0; item < C_ITEMS; item++) {
          unsigned long long k4[4];
          uint32_t m = 0;
          const unsigned mask = nms_strip(item, w, h, x0, y0, eig.data(), k4, m);
          for (int i = 0; i < 4; i++)
            if (mask >> i & 1u) got.push_back(k4[i]);
          gmax = m > gmax ? m : gmax;
        }
        for (int i = 0; i < TH * TW; i++) {
          const int r = i / TW, c = i - r * TW, x = x0 + c, y = y0 + r;
          if (x >= w || y >= h) continue;
          const float* e = eig.data() + (r + 1) * OW + (c + 1);
          const float v = e[0];
          const uint32_t ev = ordered_bits(v);
          wmax = ev > wmax ? ev : wmax;
          if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) continue;
          if (!(v > 0.f)) continue;
          bool ismax = true;
          for (int j = -1; j <= 1; j++)
            for (int k = -1; k <= 1; k++)
              if (e[j * OW + k] > v) ismax = false;
          if (ismax) want.push_back(~(((unsigned long long)ev << 32) | (unsigned)(y * w + x)));
        }
        std::sort(got.begin(), got.end());
        std::sort(want.begin(), want.end());
        keys_total += (long long)want.size();
        if (got != want || gmax != wmax) {
          if (bad < 5) std::fprintf(stderr, "tile (%d,%d): %zu keys vs %zu, max %u vs %u\n", x0, y0, got.size(), want.size(), gmax, wmax);
          bad++;
        }
      }
      for (int r = 0; r < OH; r++)
        for (int c = 0; c < OW; c++) {
          const int x = x0 - HALO + c, y = y0 - HALO + r;
          if (x < 0 || y < 0 || x >= w || y >= h) continue;
          checked++;
          if (std::memcmp(&eig[(size_t)r * OW + c], &ref[(size_t)y * w + x], 4) != 0) {
            if (bad < 5) std::fprintf(stderr, "mismatch at (%d,%d): %.9g vs %.9g\n", x, y, eig[(size_t)r * OW + c], ref[(size_t)y * w + x]);
            bad++;
          }
        }
    }
  std::printf("checked %lld responses and %lld candidate keys, %lld differ\n", checked, keys_total, bad);
  return bad ? 1 : 0;
}
