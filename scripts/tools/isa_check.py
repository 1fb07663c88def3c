#!/usr/bin/env python3
"""Build gate: scan the gfx950 ISA of every object of the library for a code-generation hazard of this toolchain (ROCm 7.2 LLVM).

Found in round 6: when the register allocator spills a VGPR that is live across a divergent region, it may place the
RELOAD at the head of the join block, BEFORE the `s_or_b64 exec, exec, s[..]` that restores the lanes which sat the region out:

        scratch_load_dword v3, off, off offset:40      ; executes under the region's (possibly EMPTY) exec mask
        s_or_b64 exec, exec, s[6:7]                    ; ... the lanes come back only here

A loop that every lane leaves at once is left with exec = 0, so the reload loads NOTHING and the register keeps whatever the loop
put there.  In hnsw_search_kernel<f32, cosine, 24 chunks, LDS beam, bitset, four waves> that register was a helper wave's index: its
share of a hop's rows was computed from garbage, rows went unevaluated, the walker read stale distances (the "keys that are not
numbers" hunt of round 5: there were none -- only this).  The source cannot express "reload after the restore", so the library is
kept free of the pattern: wave-role values are made provably uniform (scalar branches, no exec games), register budgets of the
multi-wave kernels leave no spills, and this script fails the build if the pattern shows up anywhere again.

usage: isa_check.py <objdir> [--spills]     exit code 1 when the pattern is found
"""
import glob, os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_object(obj, tmp):
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    b = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, b)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", b], capture_output=True, cwd=tmp)
    dev = glob.glob(os.path.join(tmp, "*gfx950*"))
    return dev[0] if dev else None


def short(name):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    m = re.search(r"([\w:]+<[^()]*>)\(", dn)
    return (m.group(1) if m else dn)[:110]


def scan(objdir, want_spills=False):
    bad, spills = [], []
    tmp = tempfile.mkdtemp(prefix="kdb_isa_")
    try:
        for obj in sorted(glob.glob(os.path.join(objdir, "*.o"))):
            dev = device_object(obj, os.path.join(tmp, "x"))
            if not dev:
                continue
            lines = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", dev], capture_output=True, text=True).stdout.split("\n")
            cur, hits = None, {}
            for i, l in enumerate(lines):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
                if m:
                    cur = m.group(1)
                    continue
                if l.lstrip().startswith("scratch_load"):
                    j = i + 1
                    while j < len(lines) and re.match(r"\s*(scratch_load|s_waitcnt|s_nop)", lines[j]):
                        j += 1
                    if j < len(lines) and re.match(r"\s*s_or_b64 exec, exec,", lines[j]):
                        hits[cur] = hits.get(cur, 0) + 1
            bad += [(os.path.basename(obj), short(k), v) for k, v in hits.items()]
            if want_spills:
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev], capture_output=True, text=True).stdout
                for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", e) or [None, None])[1]
                    if int(g("vgpr_spill_count") or 0) > 0:
                        spills.append((os.path.basename(obj), short(g("name")), int(g("vgpr_count")), int(g("vgpr_spill_count"))))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return bad, spills


if __name__ == "__main__":
    objdir = sys.argv[1]
    bad, spills = scan(objdir, "--spills" in sys.argv)
    for o, k, nv, ns in spills:
        print(f"spill  {o:24s} {k}: {ns} VGPRs spilled of a budget of {nv}")
    for o, k, v in bad:
        print(f"HAZARD {o:24s} {k}: {v} spill reload(s) placed before the exec restore of a join block")
    print(f"isa_check: {len(bad)} kernel(s) with the reload-before-exec-restore pattern" + (f", {len(spills)} kernel(s) spill VGPRs" if "--spills" in sys.argv else ""))
    sys.exit(1 if bad else 0)
