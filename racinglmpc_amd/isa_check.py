"""Build-time guard against a compiler fault found in round 5 (ROCm 7.2 / clang 22, gfx950).

The fault.  The register allocator splits the live range of a vector register that is live across a per-lane `if / else` and places the split's copies
(v_accvgpr_write / v_mov / scratch stores) at the TOP of the flow block that joins the two sides -- in front of the `s_or_saveexec_b64` / `s_or_b64 exec, exec, s[..]`
that restores the lanes of the other side.  The copies then execute under the mask of the `then` side only: the other lanes' copy is never made, and when the value
is copied back later (full mask) those lanes receive whatever the target register held.  Seen in lmpc_solve_kernel<40, 48, false, true> (256 VGPRs + 190 AGPRs):
the loop-invariant register arrays c_r / qsel_r came back as +inf in lanes 48..63 after the first Newton iteration, every problem of the first pass ended
LMPC_ST_NUMERIC and was rescued by the retry kernel -- "12.4 instead of 11.0 iterations and a third of the speed with every certificate green" (round 4), or, when
the retry kernel was hit as well, "every problem at the iteration limit".  Which builds are hit is decided by register allocation, i.e. by unrelated edits.
(The block prologue the allocator must stay behind is found by scanning from the block's first instruction; here an SGPR copy `s_mov_b64 s[40:41], s[18:19]`
sits in front of the `s_or_saveexec_b64`, the scan stops there, and the copies go in front of both.)

The check.  In a correct compilation nothing per-lane stands between a block label and the exec-widening instruction that opens the block: only scalar
instructions and v_readlane / v_writelane (mask reloads, which ignore EXEC).  This script disassembles the gfx950 code objects of a shared library (or reads a .s
file) and reports every block where a vector / LDS / memory instruction precedes the block's `s_or_b64 exec` / `s_or_saveexec_b64`.

racinglmpc_amd.build runs it on every library and variant it produces and recompiles with a different (semantically neutral) code-generation option
when a block is reported; tests/test_isa_check.py runs it on whatever is in the tree.

    python -m racinglmpc_amd.isa_check racinglmpc_amd/liblmpc_hip.so [more .so / .s ...]        exit status 1 if anything is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")       # (override next to HIPCC)
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WIDEN = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64\s|s_xor_b64\s+exec,\s*exec,|s_mov_b64\s+exec,\s*s)")
LANE_FREE = re.compile(r"^\s*(s_|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32|;|$)")
PER_LANE = re.compile(r"^\s*(v_|ds_|global_|flat_|scratch_|buffer_)")


def code_objects(path):
    """gfx9xx ELF images inside the clang offload bundle(s) of a host shared object."""
    data = open(path, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + ts].decode(); off += ts
            if "amdgcn" in triple and es > 0:
                out.append((triple, data[pos + eo:pos + eo + es]))
        pos = data.find(MAGIC, pos + 1)
    return out


def disassemble(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image); f.flush()
        return subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout


EXEC_WRITE = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_(or|and|andn2|xor|orn2|nand|nor|xnor)_saveexec_b64\s")
SKIP = re.compile(r"^\s*s_cbranch_execz\s+<?([\w.$]+)")


def scan(text):
    """[(function, block label, offending instruction, mask instruction)].  A block that is the target of an `s_cbranch_execz` is entered with EXEC = 0 on the
    skipping path and with the narrow mask of the region on the falling-through one: whatever per-lane instruction stands in it ahead of the first instruction
    that writes EXEC belongs to neither -- in a correct compilation there is none."""
    lines = []
    for raw in text.splitlines():
        line = raw.split("//")[0].split(";")[0].rstrip()
        lines.append(line)
    targets = set()
    for line in lines:
        m = SKIP.match(line)
        if m:
            targets.add(m.group(1).split("+")[0])
    hits = []
    func = "?"; label = None; pending = []
    for line in lines:
        st = line.strip()
        m = re.match(r"^([0-9a-f]+ )?<?([A-Za-z_.$][\w.$]*)>?:$", st)
        if m:
            name = m.group(2)
            if not (name.startswith(".L") or name.startswith("L")):
                func = name
            label = name if name in targets else None
            pending = []
            continue
        if label is None or not st or st.startswith(".") or st.startswith(";"):
            continue
        if EXEC_WRITE.match(st):
            if WIDEN.match(st):                       # the block's own prologue: lanes come back here
                for p_ in pending:
                    hits.append((func, label, p_, st))
            label = None; pending = []               # (a narrowing write opens a new region: the block had no prologue to stay behind)
        elif PER_LANE.match(st) and not LANE_FREE.match(st):
            pending.append(st)
        elif st.startswith("s_cbranch") or st.startswith("s_branch") or st.startswith("s_endpgm") or st.startswith("s_setpc"):
            label = None; pending = []               # the block ends without touching EXEC: not a flow block of the kind looked for
    return hits


def to_labelled(text):
    """llvm-objdump output (no labels: branch operands are offsets, the target is printed as <function+0xoff> behind the encoding) -> the labelled form scan() reads."""
    ins = []                                          # (address, function, text, target address or None)
    func = "?"; fstart = {}
    for raw in text.splitlines():
        m = re.match(r"^(?:[0-9a-fA-F]+\s+)?<([^>]+)>:$", raw.strip())
        if m:
            func = m.group(1); continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(<([^>+]+)(\+0x([0-9a-fA-F]+))?>)?\s*$", raw)
        if not m:
            continue
        addr = int(m.group(2), 16)
        fstart.setdefault(func, addr)
        tgt = None
        if m.group(4) is not None and m.group(1).startswith(("s_cbranch", "s_branch")):
            tgt = (m.group(4), int(m.group(6), 16) if m.group(6) else 0)
        ins.append((addr, func, m.group(1), tgt))
    labels = {}
    for addr, func, t, tgt in ins:
        if tgt is not None and tgt[0] in fstart:
            labels[fstart[tgt[0]] + tgt[1]] = ".L%x" % (fstart[tgt[0]] + tgt[1])
    out = []; cur = None
    for addr, func, t, tgt in ins:
        if func != cur:
            out.append("%s:" % func); cur = func
        if addr in labels:
            out.append("%s:" % labels[addr])
        if tgt is not None and tgt[0] in fstart:
            t = re.sub(r"\s+\S+$", " " + labels[fstart[tgt[0]] + tgt[1]], t)
        out.append("\t" + t)
    return "\n".join(out)


class IsaCheckError(RuntimeError):
    """The check could not look at the code (no code object, nothing parsed): NOT a clean result."""


def stats(text):
    """(instructions, s_cbranch_execz branches with a resolved label) of a labelled listing: what a scan of it can have seen."""
    n_ins = 0; n_skip = 0
    for raw in text.splitlines():
        st = raw.split("//")[0].split(";")[0].strip()
        if not st or st.endswith(":") or st.startswith("."):
            continue
        n_ins += 1
        if SKIP.match(st):
            n_skip += 1
    return n_ins, n_skip


def check_report(path):
    """(hits, report).  FAILS CLOSED (round 6): raises IsaCheckError unless every input yields at least one gfx9 code object, a non-zero number of parsed
    instructions and at least one `s_cbranch_execz` with a resolved target per image -- a compressed offload bundle, a change of llvm-objdump's output format or a
    missing tool must not read as "clean" (every solve kernel here has exec-mask regions, so an image without one was not parsed)."""
    texts = []
    if path.endswith(".s"):
        texts.append(("listing", open(path).read()))
    else:
        if not os.path.exists(OBJDUMP):
            raise IsaCheckError("isa_check: %s not found (set OBJDUMP)" % OBJDUMP)
        objs = code_objects(path)
        if not objs:
            raise IsaCheckError("isa_check: no amdgcn code object found in %s (compressed or unknown offload bundle?)" % path)
        for triple, img in objs:
            texts.append((triple, to_labelled(disassemble(img))))
    hits = []; images = []
    for triple, t in texts:
        n_ins, n_skip = stats(t)
        if n_ins == 0 or n_skip == 0:
            raise IsaCheckError("isa_check: %s / %s: %d instructions and %d s_cbranch_execz parsed -- the disassembly was not understood" % (path, triple, n_ins, n_skip))
        images.append({"image": triple, "instructions": n_ins, "execz_branches": n_skip})
        hits += scan(t)
    return hits, images


def check(path):
    return check_report(path)[0]


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        h = check(p)
        print("%s: %d block(s) with per-lane instructions ahead of the exec-widening prologue" % (p, len({(f, l) for f, l, _, _ in h})))
        seen = set()
        for f, l, i, w in h:
            if (f, l) not in seen:
                seen.add((f, l))
                d = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()[:90]
                print("   %s  block %s: `%s` ... before `%s`" % (d, l, i, w))
        bad += len(h)
    sys.exit(1 if bad else 0)
