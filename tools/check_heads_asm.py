"""Audit of the heads kernel's ISA (cdna_hip_programming.md §5.7: the compiler neither counts the memory operations of an
asm statement nor pads its hazards).  Compiles ocn_amd/csrc/heads.hip to assembly and checks, for every heads_fused_kernel and
heads_nsplit_kernel (whose weight fragments are requested by asm statements and waited for by counted s_waitcnt vmcnt):

  1. no instruction reads or writes the destination of a ds_read_b128 / global_load that the counted s_waitcnt
     lgkmcnt(N) / vmcnt(N) ladders have not yet retired (a compiler copy / spill of an in-flight register would be silent
     corruption);
  2. no VALU result is an MFMA A/B operand within the next 2 wait states;
  3. no non-MFMA instruction touches an MFMA's result within 12 wait states;
  4. no scratch (private memory) access, no flat access (its counters retire out of order);
  5. M0 appears only inside asm statements (the LDS-DMA statements write it and do not hand it back).

    python tools/check_heads_asm.py [file.s]          exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ops(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None, []
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def compile_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="hdasm"), "heads.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
           f"-I{ROOT}/include", f"-I{ROOT}/ocn_amd/csrc", "-S", "--cuda-device-only", "-o", out, f"{ROOT}/ocn_amd/csrc/heads.hip"]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def audit(path):
    problems, stats = [], {}
    name, body = None, []
    funcs = {}
    for line in open(path):
        m = re.match(r"^(_Z\w*heads_(?:fused|nsplit)_kernel\w*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                funcs[name] = body
                name = None
    for fn, lines in funcs.items():
        inflight = []                      # destinations of outstanding ds_reads, oldest first (None: compiler's own LDS op)
        vm = []                            # the same for vector-memory operations (loads, stores, LDS-DMA count together)
        recent = []                        # (wait states ago, kind, written regs) of the last instructions
        n_mfma = n_read = 0
        in_asm = False
        for ln, line in enumerate(lines):
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            op, ops = split_ops(line)
            if op is None:
                continue
            if not in_asm and re.search(r"\bm0\b", line.split(";")[0]):
                problems.append(f"{fn}:{ln}: compiler code uses M0: {line.strip()}")
            if "scratch_" in op or op.startswith("buffer_") and "offen" in line and "lds" not in line:
                problems.append(f"{fn}:{ln}: scratch access: {line.strip()}")
            touched = regs(" ".join(ops))
            written = regs(ops[0]) if ops and not op.startswith(("ds_write", "global_store", "s_", "global_load_lds", "ds_read")) else set()
            if op.startswith(("ds_read", "global_load_dword")):
                written = regs(ops[0])
            # 1. in-flight LDS destinations
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", line)
                if m:
                    keep = int(m.group(1))
                    while len(inflight) > keep:
                        inflight.pop(0)
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:
                    keep = int(m.group(1))
                    while len(vm) > keep:
                        vm.pop(0)
            else:
                for dst in inflight:
                    if dst and dst & touched:
                        problems.append(f"{fn}:{ln}: touches an in-flight LDS destination: {line.strip()}")
                        break
                for dst in vm:
                    if dst and dst & touched:
                        problems.append(f"{fn}:{ln}: touches an in-flight global-load destination: {line.strip()}")
                        break
            if op.startswith(("global_load_lds", "global_store", "global_atomic", "scratch_store", "buffer_store")):
                vm.append(None)
            elif op.startswith(("global_load", "scratch_load", "buffer_load")):
                vm.append(regs(ops[0]))
            elif op.startswith("flat_"):
                problems.append(f"{fn}:{ln}: flat access (out-of-order counters): {line.strip()}")
            if op.startswith("ds_read"):
                inflight.append(regs(ops[0]))
                n_read += 1
            elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                inflight.append(None)
            # 2. / 3. wait states
            states = 1
            if op == "s_nop":
                states = int(ops[0], 0) + 1
            if op.startswith("v_mfma"):
                n_mfma += 1
                ab = regs(ops[1]) | regs(ops[2])
                d = regs(ops[0])
                for ago, kind, wr in recent:
                    if kind == "valu" and ago < 2 and wr & ab:
                        problems.append(f"{fn}:{ln}: VALU result is an MFMA operand {ago} wait states later: {line.strip()}")
            else:
                for ago, kind, wr in recent:
                    if kind == "mfma" and ago < 12 and wr & touched:
                        problems.append(f"{fn}:{ln}: touches an MFMA result {ago} wait states later: {line.strip()}")
                        break
            recent = [(ago + states, k, wr) for ago, k, wr in recent if ago + states < 16]
            if op.startswith("v_mfma"):
                recent.append((0, "mfma", regs(ops[0])))
            elif op.startswith("v_"):
                recent.append((0, "valu", written))
        stats[fn] = dict(mfma=n_mfma, ds_read=n_read)
    return problems, stats


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    problems, stats = audit(path)
    for fn, s in stats.items():
        print(f"{fn}: {s['mfma']} MFMA, {s['ds_read']} ds_read")
    for p in problems[:40]:
        print("PROBLEM", p)
    print(f"{len(problems)} problems")
    sys.exit(1 if problems or not stats else 0)
