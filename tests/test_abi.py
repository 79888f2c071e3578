"""The C-ABI library: loads without a GPU, exports every symbol include/*.h declares, struct layouts match the python
binding, and the product path fails loudly (no CPU fallback) when no HIP device is usable."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_(?:hip|host)_[a-z0-9_]+)\s*\(", text)))


def test_headers_declare_what_the_library_exports(vr):
    L = C.CDLL(vr.library_path())
    names = declared_functions("vr_hip.h") + declared_functions("vr_host.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported by libvr_hip.so"
    vr.lib()     # the python binding resolves all of them too


def test_struct_layout_matches_header(vr):
    # vr_view: 2 u32 + 15 floats + u32 = 72 bytes; vr_params: view + 4+4+4+4+12+4+4 + 6*4
    assert C.sizeof(vr.VrView) == 72
    assert C.sizeof(vr.VrParams) == 72 + 36 + 24
    assert vr.VrParams.ray_step.offset == 72 and vr.VrParams.x0.offset == 108
    assert C.sizeof(vr.VrTiming) == 32 and vr.VrTiming.kernel_ms_max.offset == 24
    from importlib import import_module
    info = import_module("volume-rendering_amd.binding").VrVolumeInfo
    # 10 u32 + 2 u64 + 3 u32 + VR_COPY_KINDS = 13 floats + 1 float = 40 + 16 + 12 + 56 = 124 bytes -> 128 (u64 fields 8-aligned at 40)
    assert info.linear_bytes.offset == 40 and info.copies.offset == 56 and info.build_ms.offset == 68 and info.upload_ms.offset == 120
    assert C.sizeof(info) == 128


def test_no_cpu_fallback(vr):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is covered by test_gpu_parity.py::test_error_conventions")
    with pytest.raises(vr.VrError) as e:
        vr.HipRenderer(0)
    assert e.value.code == 2            # VR_ERR_NO_DEVICE
    ctx = C.c_void_p()
    assert vr.lib().vr_hip_create(0, C.byref(ctx)) == 2 and not ctx
    out = (C.c_uint8 * 64)()
    v = vr.benchmark_view(4, 4, 0)
    assert vr.lib().vr_host_render_frame(0, 1, C.byref(v), out) == 1   # the C++ mirror reports failure the reference's way


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "volume-rendering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "vr_oracle" not in text.replace("oracle/vr_oracle.c", "") and "libvolr_ref" not in text, os.path.join(dirpath, f)


def test_hot_kernels_keep_eight_waves_per_simd(vr):
    """The ray-march variants the product launches by default (bricked layout, table addressing) must stay within 80 SGPRs and
    64 VGPRs: on gfx950 a SIMD then holds 8 of their waves; 81-96 SGPRs silently drop that to 7, i.e. from 4 to 3 resident
    workgroups per CU (measured: +3 % frame time).  Figures come from the build's -Rpass-analysis log."""
    import subprocess
    csrc = os.path.join(ROOT, "volume-rendering_amd", "csrc")
    log = os.path.join(csrc, "resource_usage.log")
    if not os.path.exists(log):
        subprocess.check_call(["make", "-B", "-C", csrc])
    text = open(log).read()
    found = 0
    for m in re.finditer(r"Function Name: (\S*raymarch_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)E\S*).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?"
                         r"ScratchSize \[bytes/lane\]: (\d+)", text, flags=re.S):
        sampling, bpv, addr, layout = (int(m.group(i)) for i in (2, 3, 4, 5))
        sgprs, vgprs, scratch = int(m.group(6)), int(m.group(7)), int(m.group(8))
        assert scratch == 0, (m.group(1), "spills to scratch")
        if layout in (1, 2, 3, 4, 5, 6) and addr in (0, 1):     # quad bricks, run bricks (z / y), voxel bricks, oct bricks, both run copies per tile
            found += 1
            assert sgprs <= 80 and vgprs <= 64, (m.group(1), sgprs, vgprs)
    # quad bricks {NEAREST, TRILINEAR, Q8} x {u8, u16} x {32-bit, 64-bit z tables} = 12, voxel bricks (NEAREST) x 2 x 2 = 4,
    # run bricks {z, y, both per tile} x {TRILINEAR, Q8} (u8, 32-bit) = 6, oct bricks {TRILINEAR, Q8} (u16) x {32-bit, 64-bit z tables} = 4
    assert found == 26, found
    # the column march (colmarch_kernel<SAMPLING, axis, FLIPS>): {TRILINEAR, Q8} x {x, y, z} x {with, without the flip logic} = 12 kernels, none
    # may spill — a spilled scalar or vector register is reloaded inside the window loop behind an s_waitcnt that drains the prefetch
    # pipeline, and a build that was FORCED to 8 waves by spilling faulted on the GPU (round 4)
    found = 0
    for m in re.finditer(r"Function Name: (\S*colmarch_(?:nearest_kernelILi\d|kernelILi\dELi\d)ELb[01]E\S*).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                         r"SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", text, flags=re.S):
        sgprs, vgprs, scratch, sspill, vspill = (int(m.group(i)) for i in (2, 3, 4, 5, 6))
        found += 1
        assert scratch == 0 and sspill == 0 and vspill == 0, (m.group(1), "spills", scratch, sspill, vspill)
        assert sgprs <= 80 and vgprs <= 64, (m.group(1), sgprs, vgprs)
    assert found == 12 + 6, found                   # + colmarch_nearest_kernel<axis, FLIPS>


def _disassemble_gfx950(lib_path, tmp_path):
    """The gfx950 code objects embedded in the library, disassembled with the image's llvm-objdump: {symbol: [instruction lines]}."""
    import glob
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    work = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(lib_path, work)
    subprocess.check_call([objdump, "--offloading", work], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    funcs, cur = {}, None
    for bundle in sorted(glob.glob(work + ".*gfx950")):
        text = subprocess.check_output([objdump, "-d", "--mcpu=gfx950", "--no-show-raw-insn", bundle], text=True, stderr=subprocess.DEVNULL)
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = funcs.setdefault(m.group(1), [])
            elif cur is not None and line.startswith("\t") or (cur is not None and re.match(r"^\s+[a-z_0-9]+", line)):
                ins = line.split("//")[0].strip()
                if ins:
                    cur.append(ins)
    return funcs


def _vgprs(operand_text):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", operand_text):
        if m.group(3) is not None:
            regs.add(int(m.group(3)))
        else:
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def test_no_instruction_touches_a_gather_in_flight(vr, tmp_path):
    """the software-pipelined march issues its gathers through inline asm and waits with hand-counted
    `s_waitcnt vmcnt(N)`; that is only correct while the compiler never reads, copies or overwrites a destination register between
    the load and the wait that covers it.  This walks the disassembly of the hot ray-march variants IN THE BUILT LIBRARY in program
    order: every vector-memory load enters a queue with its destination registers, `s_waitcnt vmcnt(N)` retires all but the N
    youngest, and no other instruction may name a register of a load still in the queue.  (Program order is exact inside the
    straight-line loop body, where the risk is; across conditional branches it is a conservative approximation of the hardware rule,
    and a block that is only reached by a jump starts with an empty queue.)"""
    funcs = _disassemble_gfx950(vr.library_path(), tmp_path)
    checked = 0
    for name, lines in funcs.items():
        m = re.search(r"raymarch_kernelILi(\d)ELi1ELi0ELi(\d)E", name)
        column = re.search(r"colmarch_(nearest_)?kernelILi\d", name) is not None      # the column marches: one managed 16-byte gather per window
        if not column and (not m or int(m.group(2)) not in (1, 2, 3, 4, 6)):
            continue
        checked += 1
        # a register that appears in the function ONLY as the destination of loads is a sink (the column march warms the caches ahead of its
        # event windows with loads whose result nobody reads): two such loads in flight into it cannot hurt anybody
        elsewhere, load_dst = set(), set()
        for ins in lines:
            parts = ins.split(None, 1)
            op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            if re.match(r"(global|flat|buffer|scratch)_load", op):
                load_dst |= _vgprs(rest.split(",")[0])
            if re.match(r"(global|flat|buffer|scratch|ds)_(store|write|atomic)|v_cmp|v_readlane|v_readfirstlane", op):
                elsewhere |= _vgprs(rest)                      # no vector destination: every operand is read
            else:
                elsewhere |= _vgprs(",".join(rest.split(",")[1:]))     # everything but the destination
        sinks = load_dst - elsewhere
        assert len(sinks) <= 1, (name, sinks)
        inflight, loads, waits = [], 0, 0
        for ins in lines:
            parts = ins.split(None, 1)
            op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            if op == "s_waitcnt":
                w = re.search(r"vmcnt\((\d+)\)", rest)
                if w:
                    n = int(w.group(1))
                    inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
                    if n == 0:
                        inflight = []
                    waits += 1
                continue
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                # what follows is not reached by falling through: a block the compiler moved out of line (the dense path of a window step sits
                # behind the loop in the binary, i.e. AFTER the next step's gather in program order, but executes before it).  Its
                # predecessors are unknown to this walk: start it with an empty queue rather than with a false alarm.
                inflight = []
                continue
            regs = _vgprs(rest) - sinks
            busy = set().union(*inflight) if inflight else set()
            assert not (regs & busy), f"{name}: `{ins}` names v{sorted(regs & busy)} while a load into it is in flight"
            if re.match(r"(global|flat|buffer|scratch)_load", op):
                inflight.append(_vgprs(rest.split(",")[0]))
                loads += 1
            elif re.match(r"(global|flat|buffer|scratch)_(store|atomic)", op):
                inflight.append(set())                      # shares the counter; has no destination to protect
        assert (loads >= 20 or column and loads >= 8) and waits >= 10, (name, loads, waits)
    assert checked >= 10 + 12 + 6, checked
