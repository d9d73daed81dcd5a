"""Writes median_keys_dbg.inc: a copy of median_keys (vip_amd/csrc/collapse.hip) that returns an intermediate after stage STOP,
for probe2.hip (which stage of the selection first differs beside a co-runner).   python tools/hunt/gen_dbg.py"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "..", "vip_amd", "csrc", "collapse.hip")).read()
a = src.index('template <int RPL>\n__device__ __forceinline__ void median_keys(')
b = src.index('constexpr int HIST_WORDS')
fn = src[a:b]
fn = fn.replace('template <int RPL>\n__device__ __forceinline__ void median_keys(', 'template <int RPL, int STOP>\n__device__ __forceinline__ void median_keys_dbg(')
fn = fn.replace('unsigned& klow, unsigned& khigh) {', 'unsigned& klow, unsigned& khigh, unsigned& dbg) {')


def rep(x, y):
    global fn
    assert x in fn, x
    fn = fn.replace(x, y, 1)


rep('  int rank = k;                                  // rank of the wanted key',
    '  if (STOP == 1) { dbg = lo ^ (hi * 3u); return; }\n  int rank = k;                                  // rank of the wanted key')
rep('    const unsigned s4 = h.x + h.y + h.z + h.w;', '''    if (STOP == 3) {
      dbg = h.x + 3 * h.y + 5 * h.z + 7 * h.w;
      unsigned t_ = 0;
_Pragma("unroll")
      for (int r = 0; r < RPL; ++r) t_ = t_ * 257u + (unsigned)bin[r];
      klow = t_; khigh = __float_as_uint(scale) ^ lo ^ (hi << 1);
      return;
    }
    const unsigned s4 = h.x + h.y + h.z + h.w;''')
rep('    const unsigned long long above = __ballot(incl > (unsigned)rank);',
    '    if (STOP == 4) { dbg = incl; return; }\n    const unsigned long long above = __ballot(incl > (unsigned)rank);')
rep('    const int bstar = 4 * L + j;\n    rank = (int)rem;',
    '    const int bstar = 4 * L + j;\n    rank = (int)rem;\n    if (STOP == 5) { dbg = (unsigned)L | ((unsigned)bstar << 8) | (c << 18) | (rem << 26); return; }')
rep('      const unsigned cand = (unsigned)lane < c ? hist[lane] : 0xffffffffu;', '''      const unsigned cand = (unsigned)lane < c ? hist[lane] : 0xffffffffu;
      if (STOP == 6) { unsigned x = cand; for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64); dbg = x; return; }''')
rep('      const bool mine = (unsigned)lane < c;', '      if (STOP == 7) { dbg = (unsigned)less; return; }\n      const bool mine = (unsigned)lane < c;')
open(os.path.join(HERE, "median_keys_dbg.inc"), "w").write(fn)
print("wrote median_keys_dbg.inc")
