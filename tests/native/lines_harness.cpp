// CPU harness for facet_amd/csrc/lines_host.cpp (built with -fsanitize=address,undefined by tests/test_lines_host.py).
// in : int32 h, w, threshold, min_len, max_gap, max_lines, threads, n ; n * h * w bytes (NMS maps: 2 / 0 / 1)
// out: n * h * w bytes (edges 0 / 255) ; n int32 counts ; n * max_lines * 4 int32 segments
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lines_host.h"

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int hdr[8];
  if (fread(hdr, sizeof(int), 8, f) != 8) return 4;
  const int h = hdr[0], w = hdr[1], thr = hdr[2], min_len = hdr[3], max_gap = hdr[4], max_lines = hdr[5], threads = hdr[6], n = hdr[7];
  std::vector<uint8_t> maps((size_t)n * h * w);
  if (fread(maps.data(), 1, maps.size(), f) != maps.size()) return 5;
  fclose(f);
  std::vector<int> lines((size_t)n * max_lines * 4, -1), counts(n, -1);
  fe::lines_host_stage(maps.data(), n, h, w, thr, min_len, max_gap, max_lines, lines.data(), counts.data(), threads);
  f = fopen(argv[2], "wb");
  if (!f) return 6;
  fwrite(maps.data(), 1, maps.size(), f);
  fwrite(counts.data(), sizeof(int), counts.size(), f);
  fwrite(lines.data(), sizeof(int), lines.size(), f);
  fclose(f);
  return 0;
}
