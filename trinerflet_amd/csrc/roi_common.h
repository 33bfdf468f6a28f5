// roi_common.h -- region of interest of the plane grid shared by the IDWT, layout and scatter kernels.
#pragma once
#include <stdint.h>

// Per plane (slice s of a depthwise kernel belongs to plane (s0 + s) / spp): origin (ox, oy) and common size rw x rh in
// texels of the grid the kernel's fine side lives on, all multiples of 64.  rw == 0 means the whole plane.  With a
// ROI the fine-side array (IDWT forward output, plane gradient, adjoint input) is COMPACT: [S][rh][rw].
// TrainStep derives the ROI from the occupancy grid: texels no sample can touch are neither rebuilt nor
// transposed, and their (identically zero) gradient is neither stored nor read.
struct Roi {
  int ox[3], oy[3];
  int rw, rh, spp, s0;
  int strided;   // 1: the array keeps its full row stride / slice size and only the window is valid (zero outside)
};

// host array {ox0,ox1,ox2, oy0,oy1,oy2, rw, rh, spp, s0} (NULL = whole plane) -> Roi; false if malformed.
// spp = slices (channels) per plane, s0 = global index of the first slice handed to a depthwise call (a rank
// that owns slices [s0, s1) of the 3*spp passes S = s1 - s0 and this s0).
static inline bool make_roi(const int32_t* r, uint32_t S, uint32_t fine_n, Roi& roi, int align = 64) {
  roi = Roi{};
  if (r == nullptr) return true;
  for (int p = 0; p < 3; p++) { roi.ox[p] = r[p]; roi.oy[p] = r[3 + p]; }
  roi.rw = r[6]; roi.rh = r[7]; roi.spp = r[8]; roi.s0 = r[9];
  if (roi.spp <= 0 || roi.s0 < 0 || (uint64_t)roi.s0 + S > 3ull * (uint32_t)roi.spp) return false;
  if (roi.rw <= 0 || roi.rh <= 0 || roi.rw % align || roi.rh % align) return false;
  for (int p = 0; p < 3; p++)
    if (roi.ox[p] < 0 || roi.oy[p] < 0 || roi.ox[p] % align || roi.oy[p] % align || roi.ox[p] + roi.rw > (int)fine_n ||
        roi.oy[p] + roi.rh > (int)fine_n)
      return false;
  return true;
}
