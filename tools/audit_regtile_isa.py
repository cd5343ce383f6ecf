#!/usr/bin/env python3
"""Audit of the hand-issued mail loads of lbm_regtile's asynchronous loop (lbm_regtile.hip.h, "the asynchronous loop").

An inline-asm load's destination registers belong to the compiler as soon as the asm statement ends, although the data
lands hundreds of cycles later: any compiler instruction that reads, copies or overwrites them before the wait statement
that retires them (marked `; retire v[a:b] ...` in the asm text) sees -- or destroys -- data that is not there yet
(cdna_hip_programming.md 5.7, item 1).  This walks the ISA of every lbm_regtile<R, MODE | kRegAsync> instantiation from each
asm `buffer_load_dwordx4` along every path (branches followed both ways, loops once round) until the retiring statement, and
reports every instruction outside asm statements that touches the destination registers on the way; also scratch use and
AGPR traffic.  Exit status 1 on any finding.

    python tools/audit_regtile_isa.py [file.s]      (default: compiles advanced-hpc-lbm_amd/csrc/lbm_api.hip -S for gfx950)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="lbm_isa_"), "lbm_api.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-function",
                    "-Wno-align-mismatch", "-Wno-pass-failed", "--offload-device-only", "-S",
                    os.path.join(ROOT, "advanced-hpc-lbm_amd", "csrc", "lbm_api.hip"), "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def regset(text):
    s = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        s |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        s.add(int(a))
    return s


def audit_kernel(name, body):
    # instructions with flags: in_asm, label targets
    ins, labels, in_asm = [], {}, False
    for raw in body:
        ls = raw.strip()
        if ls.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ls.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ls or ls.startswith(";") or ls.startswith("."):
            if re.match(r"^\.?[A-Za-z_][\w.$]*:", ls):
                labels[ls.split(":")[0]] = len(ins)
            continue
        if re.match(r"^[A-Za-z_.][\w.$]*:", ls):
            labels[ls.split(":")[0]] = len(ins)
            continue
        ins.append((in_asm, ls))
    findings = []
    nloads = 0
    for p, (ia, ls) in enumerate(ins):
        if not (ia and ls.startswith("buffer_load_dwordx4")):
            continue
        nloads += 1
        dest = regset(ls.split()[1].rstrip(","))
        seen, stack, retired_somewhere = set(), [p + 1], False
        while stack:
            q = stack.pop()
            while q < len(ins) and q not in seen:
                seen.add(q)
                ja, ms = ins[q]
                if ja and "retire" in ms and dest <= regset(ms.split(";", 1)[1]):
                    retired_somewhere = True
                    break
                if ja and ms.startswith("buffer_load_dwordx4") and regset(ms.split()[1].rstrip(",")) == dest:
                    break                      # fetched again into the same registers (slow path): that load is audited on its own
                if not ja and (regset(ms) & dest):
                    findings.append(f"{name}: load #{p} into v{sorted(dest)}: touched before its wait by `{ms}`")
                m = re.match(r"s_(c?branch\w*)\s+(\S+)", ms)
                if m and m.group(2) in labels:
                    stack.append(labels[m.group(2)])
                    if m.group(1) == "branch":
                        break                  # unconditional
                if ms.startswith("s_endpgm"):
                    break
                q += 1
        if not retired_somewhere:
            findings.append(f"{name}: load #{p} into v{sorted(dest)}: no retiring wait found on any path")
    # the asm loads and stores carry no wait states of their own: their descriptor and scalar offset must not have been
    # written by a VECTOR instruction (v_readfirstlane, v_readlane, v_cmp ... to SGPRs) within the five instructions in front
    def sregs(text):
        s = set()
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
            s |= set(range(int(a), int(b) + 1))
        for a in re.findall(r"\bs(\d+)\b", text):
            s.add(int(a))
        return s
    for p, (ia, ls) in enumerate(ins):
        if ia and (ls.startswith("buffer_load_dwordx4") or ls.startswith("buffer_store_dwordx4")):
            need = sregs(ls)
            for q in range(max(0, p - 5), p):
                ja, ms = ins[q]
                if ms.startswith("v_") and (sregs(ms.split(",")[0]) & need or ("vcc" in ms.split(",")[0] and "vcc" in ls)):
                    findings.append(f"{name}: `{ls}` reads a scalar register that `{ms}` wrote {p - q} instruction(s) earlier")
    agpr = [ls for ia, ls in ins if not ia and "accvgpr" in ls]
    if agpr:
        findings.append(f"{name}: compiler AGPR traffic: {agpr[0]} (+{len(agpr) - 1} more)")
    return nloads, findings


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else device_asm()
    lines = open(path).read().split("\n")
    total, allf, kernels = 0, [], 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN3lbm(?:11lbm_regtile|17lbm_regtile_slabs)ILi(\d+)ELi(\d+)EEEv(?:NS_11RegTileArgsE|PKNS_11RegTileArgsE)):", lines[i])
        if m and (int(m.group(3)) & 4096) and int(m.group(2)) > 1:
            j = i
            while j < len(lines) and not lines[j].strip().startswith(".end_amdhsa_kernel"):
                j += 1
            body = lines[i:j]
            scratch = [ln for ln in body if "private_segment_fixed_size" in ln]
            kname = ("lbm_regtile_slabs" if "slabs" in m.group(1) else "lbm_regtile") + f"<{m.group(2)}, {m.group(3)}>"
            n, f = audit_kernel(kname, body)
            if scratch and not scratch[0].strip().endswith(" 0"):
                f.append(f"{kname}: scratch in use: {scratch[0].strip()}")
            print(f"{kname}: {n} asm loads audited, {len(f)} finding(s)")
            total += n
            allf += f
            kernels += 1
            i = j
        i += 1
    for f in allf[:40]:
        print("  ", f)
    if kernels == 0 or total == 0:
        print("no asynchronous lbm_regtile instantiation found in", path)
        return 1
    return 1 if allf else 0


if __name__ == "__main__":
    sys.exit(main())
