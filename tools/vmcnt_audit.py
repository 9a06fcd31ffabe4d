#!/usr/bin/env python3
"""Build-time audit of counted vector-memory waits.

Round 3 met a wrong-operand failure in bps_mfma_kernel that disappeared when a compiler-counted `s_waitcnt vmcnt(N > 0)`
over loads of MIXED widths / address forms (one 12-byte `global_load_dwordx3 v, v[a:b], off` among 16-byte
`global_load_dwordx4 v, v, s[a:b]`) was replaced by `vmcnt(0)`.  The mechanism was never pinned down, so
the pattern itself is what this script looks for, in EVERY kernel of the library:

    for each kernel of each csrc/*.hip (compiled to gfx950 assembly with build.sh's flags), walk the instruction stream,
    keep the queue of outstanding vector-memory operations (loads, LDS-DMA, stores, atomics: they share vmcnt, in issue
    order), and at every `s_waitcnt vmcnt(N)` with N > 0 that retires at least one LOAD report the wait if the loads
    outstanding at that point are not all of one class (class = width x address form x to-LDS).

The queue is followed over the kernel's control-flow graph (basic blocks split at labels and branches) to a fixed point:
queues that meet at a block are joined entry by entry, counted from the newest operation, so every path is covered — the
join over-approximates (it can report a mix that no single path has, never the reverse).  Output: one line per kernel with the mixes found; exit code 1
if a kernel NOT on the justified list below has one.  `python3 tools/vmcnt_audit.py > profiles/r04_vmcnt_audit.txt`.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dex_gan_grasp_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFILT = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(p)), "c++filt")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function"]
EXTRA = {  # as in csrc/build.sh
    "gemm_x5.hip": ["-fno-slp-vectorize"],
    "gemm_x4.hip": ["-fno-slp-vectorize"],
    "bps_encode.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"],
}

# Kernels whose mixed counted waits were read by hand and are accepted, with the reason (substring of the demangled name).
JUSTIFIED = {
    "gemm_x5_kernel<true, true, true, ":
        "compiler spill reloads only (scratch_load_dword / dwordx2 of 8 set-up values, private segment), in the prologue, outside "
        "the k loop; the kernel's own waits on its LDS-DMA pieces (two widths: 16-byte operand pieces, 4-byte affine pieces) are "
        "all vmcnt(0) — gemm_x5.hip has no counted wait of its own",
    "fc2_stage_kernel<":
        "one counted wait (for bn1's gamma) at the join behind the train / eval branch of the statistics prologue, with the 16-byte "
        "operand loads in flight only along the compiler's EXEC = 0 skip of the eval side (s_cbranch_execz): no lane ever takes "
        "it — every lane runs the train or the eval side, and both wait vmcnt(0) (explicit asm) before their first use of a "
        "loaded value, so at run time that wait finds nothing outstanding",
    "gemm_x4_kernel<false, false, false>":
        "compiler spill reloads (scratch_load_dword, private segment; 9 registers) among the loader waves' dwordx4 loads.  The TN "
        "instantiation of the >= 1024-workgroup kernel is reached by no launch of any BASELINE config (weight-gradient launches "
        "have <= 240 workgroups); it is compared with fp64 in test_gemm_bf16x3_variants[*-2-shape8] on the GPU",
}

# Accepted as a class: loads of ONE width, all returning to registers (or all to LDS), that differ only in how the
# instruction encodes the address.  Listed in the output, not counted as open.
FORM_ONLY = ("address form only: one width, one destination kind; `v, s[a:b]` (SGPR base + 32-bit VGPR offset) and `v[a:b], off` "
             "(64-bit VGPR address) are two encodings of the same naturally aligned request")


def form_only(loads):
    return len({(c.split("/")[0], c.endswith("/lds")) for c in loads}) == 1


VMEM = re.compile(r"^\s+((?:global|buffer|flat|scratch)_(?:load|store|atomic)\S*)\s+(.*)$")
WAIT = re.compile(r"^\s+s_waitcnt\s+(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z\w+):\s*; @")


def classify(mn, ops):
    is_load = "_load" in mn or ("_atomic" in mn and " sc0" in ops)   # returning atomics write registers too
    w = re.search(r"(dwordx[234]|dword|ubyte|sbyte|ushort|sshort|short_d16\w*|b\d+)", mn)
    width = w.group(1) if w else mn
    to_lds = "_lds_" in mn or ops.rstrip().endswith(" lds") or " lds " in ops
    if mn.startswith("scratch"):
        form = "scratch"   # private segment: the compiler's spill reloads
    elif mn.startswith("flat"):
        form = "flat"
    elif mn.startswith("buffer"):
        form = "buffer"
    elif re.search(r",\s*off\b", ops):
        form = "vaddr64"
    else:
        form = "saddr"
    kind = "load" if is_load else ("store" if "_store" in mn else "atomic")
    return kind, "%s/%s%s" % (width, form, "/lds" if to_lds else "")


def vmcnt_of(text):
    """N of an s_waitcnt operand string, None if vmcnt is not waited for."""
    m = re.search(r"vmcnt\((\d+)\)", text)
    if m:
        return int(m.group(1))
    t = text.strip()
    if re.fullmatch(r"0(x0+)?", t):   # `s_waitcnt 0`: everything
        return 0
    return None


TERM = re.compile(r"^\s+(s_endpgm|s_setpc_b64|s_branch|s_cbranch\S*)\b\s*(\.LBB\d+_\d+)?")


def blocks_of(lines, lo, hi):
    """Basic blocks [start, end) of one function and their successors."""
    starts = {lo}
    for i in range(lo, hi):
        if LABEL.match(lines[i]):
            starts.add(i)
        elif TERM.match(lines[i]):
            starts.add(i + 1)
    order = sorted(x for x in starts if x < hi)
    label_at = {}
    for i in range(lo, hi):
        m = LABEL.match(lines[i])
        if m:
            label_at[m.group(1)] = i
    blocks = {}
    for k, st in enumerate(order):
        en = order[k + 1] if k + 1 < len(order) else hi
        succ = [en] if en < hi else []
        m = TERM.match(lines[en - 1]) if en - 1 >= st else None
        if m:
            op, tgt = m.group(1), m.group(2)
            if op in ("s_endpgm", "s_setpc_b64"):
                succ = []
            elif op == "s_branch":
                succ = [label_at[tgt]] if tgt in label_at else []
            elif tgt in label_at:
                succ = succ + [label_at[tgt]]
        blocks[st] = (en, succ)
    return blocks


def join(a, b):
    """Elementwise union of two abstract queues, aligned at their NEWEST entry (what vmcnt(N) counts from)."""
    if len(a) < len(b):
        a, b = b, a
    pad = len(a) - len(b)
    return tuple(a[i] | (b[i - pad] if i >= pad else frozenset()) for i in range(len(a)))


def walk(lines, lo, hi, found):
    """Fixed point over the control-flow graph.  The abstract state at a program point is the queue of outstanding
    vector-memory operations, oldest first, each entry the SET of (kind, class) that some path puts at that distance from
    the newest operation; states that meet at a block are joined entry by entry.  An over-approximation of every path
    (it never misses a mix a real path has; it may report one no single path has)."""
    blocks = blocks_of(lines, lo, hi)
    state = {lo: ()}
    work = [lo]
    reported = set()
    while work:
        st = work.pop()
        en, succ = blocks[st]
        queue = list(state[st])
        for i in range(st, en):
            ln = lines[i]
            m = VMEM.match(ln)
            if m:
                queue.append(frozenset([classify(m.group(1), m.group(2))]))
                continue
            m = WAIT.match(ln)
            if m:
                n = vmcnt_of(m.group(1))
                if n is not None:
                    if n > 0 and len(queue) > n:
                        retired = queue[:len(queue) - n]
                        loads = tuple(sorted({c for e in queue for k, c in e if k == "load"}))
                        if any(k == "load" for e in retired for k, _ in e) and len(loads) > 1 and (i, n, loads) not in reported:
                            reported.add((i, n, loads))
                            found.append((i, n, loads))
                    del queue[:max(0, len(queue) - n)]
        out = tuple(queue[-63:])   # vmcnt is a 6-bit counter: a wait never counts further back
        for nx in succ:
            old = state.get(nx)
            new = out if old is None else join(old, out)
            if new != old:
                state[nx] = new
                work.append(nx)
    return False


def demangle(names):
    if not names:
        return []
    try:
        out = subprocess.run([CXXFILT] + names, capture_output=True, text=True, check=True, stdin=subprocess.DEVNULL).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def audit_file(src):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + COMMON + EXTRA.get(os.path.basename(src), []) + ["-S", "--cuda-device-only", src, "-o", asm],
                       check=True, capture_output=True, cwd=CSRC)
        lines = open(asm).read().split("\n")
    funcs = []
    for i, ln in enumerate(lines):
        m = FUNC.match(ln)
        if m:
            funcs.append([m.group(1), i, None])
        elif ln.startswith(".Lfunc_end") and funcs and funcs[-1][2] is None:
            funcs[-1][2] = i
    res = []
    for name, lo, hi in funcs:
        hi = hi or len(lines)
        found = []
        capped = walk(lines, lo, hi, found)
        n_counted = sum(1 for i in range(lo, hi) if WAIT.match(lines[i]) and (vmcnt_of(WAIT.match(lines[i]).group(1)) or 0) > 0)
        n_x3 = sum(1 for i in range(lo, hi) if re.match(r"^\s+global_load_dwordx3", lines[i]))
        n_flat = sum(1 for i in range(lo, hi) if re.match(r"^\s+flat_(load|store|atomic)", lines[i]))
        res.append((name, n_counted, n_x3, n_flat, found, capped))
    return res


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    print("# vmcnt audit: kernels with a counted s_waitcnt vmcnt(N > 0) that retires a load while loads of more than one class")
    print("# (width/address form[/lds]) are outstanding.  Columns: file | kernel | counted waits | dwordx3 loads | flat ops | mixed waits")
    for f in srcs:
        res = audit_file(os.path.join(CSRC, f))
        names = demangle([r[0] for r in res])
        for (name, n_counted, n_x3, n_flat, found, capped), dn in zip(res, names):
            short = re.sub(r"^void ", "", re.sub(r"\(.*", "", dn))
            mixes = {}
            for _, n, loads in found:
                mixes[loads] = mixes.get(loads, 0) + 1
            why = next((v for k, v in JUSTIFIED.items() if k in dn), None)
            if found and not why and all(form_only(loads) for loads in mixes):
                why = FORM_ONLY
            status = "clean" if not found else ("JUSTIFIED: " + why if why else "MIXED")
            if found and not why:
                bad += 1
            print("%-18s %-70s counted %3d  x3 %2d  flat %2d  mixed %3d  %s" % (f, short[:70], n_counted, n_x3, n_flat, len(found), status))
            for loads, cnt in sorted(mixes.items(), key=lambda kv: -kv[1]):
                print("%20s%3d x  %s" % ("", cnt, "  +  ".join(loads)))
    print("# kernels with unjustified mixed counted waits: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
