#!/usr/bin/env python3
"""Device-code lint for libvipmi: fails when an instruction form that gfx950 executes wrongly is present.

The form (measured: tools/hunt/probe4.hip, probe5.hip; vip_amd/csrc/common.h):
    v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel[src0] = 0 and op_sel[src1] = 1   (`op_sel:[0,1]`, `op_sel:[0,1,x]`)
returns a wrong low half in lanes 48..63 while another wave of the CU runs a 16-byte-operand 16x16 MFMA
(v_mfma_i32_16x16x64_i8, v_mfma_f32_16x16x32_bf16 / _f16).  The mirrored form (swizzle on src0) is exact.

    python tools/isa_lint.py [file.o | file.so ...]        default: vip_amd/csrc/*.o and vip_amd/libvipmi.so

Every gfx950 code object embedded in the files' .hip_fatbin sections is disassembled with llvm-objdump; exit status 1 and a list
of (file, kernel, instruction) on any hit.
"""
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("VIPMI_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAD = re.compile(r"\bv_pk_(?:add|mul|fma)_f32\b[^\n]*\bop_sel:\[0,1[,\]]")


def code_objects(path):
    """(name, bytes) of every gfx950 code object in the file's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path, os.devnull],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            # no device code in this object (a host-only translation unit has no .hip_fatbin section) -- or the tool failed
            sec = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", path], capture_output=True, text=True)
            if sec.returncode == 0 and ".hip_fatbin" not in sec.stdout:
                return None
            raise RuntimeError("llvm-objcopy could not extract .hip_fatbin from %s: %s" % (path, r.stderr[:300]))
        data = open(fat, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def lint_file(path):
    hits, ninstr = [], 0
    objs = code_objects(path)
    if objs is None:
        return [], None                      # host-only object: nothing to lint
    if not objs:
        raise RuntimeError("%s has a .hip_fatbin section but no readable gfx950 code object (a compressed CCOB bundle of a newer "
                           "toolchain? unbundle with clang-offload-bundler): lint cannot vouch for it" % path)
    for i, (triple, blob) in enumerate(objs):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump failed on %s (code object %d): %s" % (path, i, r.stderr[:300]))
        kernel = "?"
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
                continue
            if "\t" in line:
                ninstr += 1
                if BAD.search(line):
                    hits.append((path, kernel, " ".join(line.split("//")[0].split())))
    return hits, ninstr


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, "vip_amd", "csrc", "*.o"))) + [os.path.join(ROOT, "vip_amd", "libvipmi.so")]
    files = [f for f in files if os.path.exists(f)]
    if not files:
        print("isa_lint: nothing to check (build first)")
        return 2
    allhits, total, empty = [], 0, []
    for f in files:
        hits, n = lint_file(f)
        allhits += hits
        if n is None:
            continue
        total += n
        if n == 0:
            empty.append(f)
    # a gate that cannot see the code is no gate: a file without a readable gfx950 code object (llvm-objcopy failed, no .hip_fatbin,
    # a compressed CCOB bundle of a newer toolchain) or an implausibly small disassembly fails the build instead of passing it
    if empty:
        print("isa_lint: no gfx950 instructions found in %s -- the offload bundle could not be read (compressed bundle? "
              "missing .hip_fatbin?): lint cannot vouch for these files" % ", ".join(os.path.basename(f) for f in empty))
        return 3
    if any(f.endswith(".so") for f in files) and total < 100000:
        print("isa_lint: only %d instructions disassembled from a library build -- implausible, refusing to pass" % total)
        return 3
    if allhits:
        kernels = {}
        for path, kernel, ins in allhits:
            kernels.setdefault((os.path.basename(path), kernel), []).append(ins)
        print("isa_lint: %d instructions of the forbidden packed-FP32 form (src0 low / src1 high) in %d kernels:" % (len(allhits), len(kernels)))
        for (fn, k), ins in sorted(kernels.items()):
            print("  %s  %s  x%d   e.g. %s" % (fn, k, len(ins), ins[0]))
        return 1
    print("isa_lint: %d files, %d instructions, no forbidden packed-FP32 operand form" % (len(files), total))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
