// rot_common.h -- geometry shared by the direct and FFT derotation paths.
#pragma once
#include <hip/hip_runtime.h>

namespace vipmi {

struct RotGeom {
  int N;      // frame size (square)
  int L;      // padded canvas size (derotation.py:154-158 then cosmetics.py:210-215)
  int Le;     // even work length (L or L-1), derotation.py:583-599
  int off;    // frame occupies canvas [off, off+N): pixel N//2 sits on L//2
  int c;      // shear origin = frame_center(canvas) = L//2
  int Lc;     // rot90 maps index i -> Lc - i (pivot pixel L//2)
  int alt0;   // first occupied index of an axis reversed by the rot90 pre-step = Lc-off-N+1
};

struct RotFrame {
  double a;   // tan(d/2)
  double b;   // -sin(d)
  int q;      // quarter turns (np.rot90 count, mod 4)
  int pad;
};

// canvas'(Y, X) after np.rot90(canvas, q)  ->  source frame pixel (fy, fx) (may be out of range)
__host__ __device__ __forceinline__ void rot_src(int q, int Y, int X, const RotGeom& g, int& fy, int& fx) {
  switch (q) {
    case 1: fy = X - g.off; fx = g.Lc - Y - g.off; break;
    case 2: fy = g.Lc - Y - g.off; fx = g.Lc - X - g.off; break;
    case 3: fy = g.Lc - X - g.off; fx = Y - g.off; break;
    default: fy = Y - g.off; fx = X - g.off; break;
  }
}

}  // namespace vipmi
