// Host stage of leading-line detection (lines_host.cpp).
#pragma once
#include <cstdint>

namespace fe {
// map [h][w]: 2 = edge, 0 = edge if 8-connected to a 2, 1 = not an edge -> in place 255 / 0
void canny_hysteresis(uint8_t* map, int h, int w);
// edges [h][w] (non-zero = edge) -> first max_lines segments (x1,y1,x2,y2) into `lines`; returns the number found (may exceed max_lines)
int hough_lines_p(const uint8_t* edges, int h, int w, int threshold, int min_len, int max_gap, int max_lines, int* lines);
// maps [n][h][w] -> edges in place and, when lines != nullptr, lines [n][max_lines][4] + counts [n]; one image per host thread
void lines_host_stage(uint8_t* maps, int n, int h, int w, int threshold, int min_len, int max_gap, int max_lines, int* lines, int* counts, int threads);
}  // namespace fe
