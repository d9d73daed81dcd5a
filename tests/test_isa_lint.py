"""The built library must not contain the packed-FP32 operand form that gfx950 executes wrongly beside another wave's
16x16x64-i8 / 16x16x32-bf16 / f16 MFMA (vip_amd/csrc/common.h, VIPMI_NO_PK32; measured by tools/hunt/probe5.hip).  Runs on CPU:
the device code is disassembled from libvipmi.so."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vip_amd", "libvipmi.so")


def _lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_forbidden_form_is_recognised():
    lint = _lint()
    bad = ["v_pk_add_f32 v[42:43], v[14:15], v[20:21] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
           "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[0,0]",
           "v_pk_fma_f32 v[8:9], v[30:31], v[8:9], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
           "v_pk_fma_f32 v[8:9], v[30:31], v[8:9], v[12:13] op_sel:[0,1,1]"]
    good = ["v_pk_add_f32 v[42:43], v[14:15], v[20:21] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]",
            "v_pk_add_f32 v[0:1], v[0:1], s[2:3] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
            "v_pk_fma_f32 v[8:9], v[30:31], v[8:9], v[12:13] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]",
            "v_pk_fma_f32 v[8:9], v[30:31], v[8:9], v[12:13] op_sel:[0,0,1]",
            "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]",
            "v_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]"]
    for ins in bad:
        assert lint.BAD.search(ins), ins
    for ins in good:
        assert not lint.BAD.search(ins), ins


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvipmi.so not built")
def test_library_has_no_forbidden_packed_fp32_form():
    lint = _lint()
    objs = lint.code_objects(LIB)
    assert objs, "no gfx950 code object found in libvipmi.so"
    hits, ninstr = lint.lint_file(LIB)
    assert ninstr > 100000, "disassembly looks empty (%d instructions)" % ninstr
    assert not hits, "forbidden packed-FP32 operand form in: %s" % sorted({(k, i) for _, k, i in hits})[:5]


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvipmi.so not built")
def test_the_aggressor_instruction_is_still_where_we_think_it_is():
    """The int8 Gram product is the one kernel of the library that runs the 16x16x64 int8 MFMA: if that ever changes (a new
    16-byte-operand 16x16 MFMA somewhere else) the co-residency reasoning of DESIGN 3.4 has to be redone."""
    lint = _lint()
    import subprocess
    import tempfile
    users = set()
    for _, blob in lint.code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([os.path.join(lint.LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True,
                                 text=True).stdout
        kernel = "?"
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
            elif re.search(r"v_mfma_(i32_16x16x64_i8|f32_16x16x32_(bf16|f16))", line):
                users.add(kernel)
    assert users and all("gram_i8_kernel" in k for k in users), users


def test_lint_cannot_pass_vacuously(tmp_path):
    """a gate that cannot see the device code must fail, not pass.  A host-only object has nothing to lint; a file whose
    offload bundle cannot be read raises; a library build with an implausibly small disassembly is refused."""
    import subprocess
    lint = _lint()
    src = tmp_path / "host_only.c"
    src.write_text("int f(int x) { return x + 1; }\n")
    obj = tmp_path / "host_only.o"
    subprocess.run(["gcc", "-c", str(src), "-o", str(obj)], check=True)
    assert lint.code_objects(str(obj)) is None
    assert lint.lint_file(str(obj)) == ([], None)
    # an object that HAS a .hip_fatbin section without a readable gfx950 code object (here: garbage in the section)
    bad = tmp_path / "bad.o"
    blob = tmp_path / "blob.bin"
    blob.write_bytes(b"\0" * 256)
    r = subprocess.run([os.path.join(lint.LLVM, "llvm-objcopy"), "--add-section", ".hip_fatbin=" + str(blob), str(obj), str(bad)],
                       capture_output=True, text=True)
    if r.returncode == 0:
        with pytest.raises(RuntimeError):
            lint.lint_file(str(bad))
    # a "library" with next to no device instructions is not a pass
    fake = tmp_path / "fake.so"
    fake.write_bytes(obj.read_bytes())
    assert lint.main([str(fake)]) != 0
