#!/usr/bin/env python3
"""Find vector instructions that run with EXEC == 0 behind the exit of a divergent loop (gfx950 code objects or hipcc -S output).

Why (round 5): hipcc lowers a divergent loop to

        loop:   ...
                s_andn2_b64 exec, exec, s[done]      ; lanes that are finished leave
                s_cbranch_execnz loop
        exit:   s_or_b64 exec, exec, s[saved]        ; the lanes of the enclosing region come back

so the fall-through of the back edge is reached with EXEC == 0 and the exit block must restore EXEC before anything else.  When
the loop is the tail of an `if` whose own end is followed directly by the end of an enclosing `if`, the compiler drops the inner
restore as redundant (SILowerControlFlow, -amdgpu-remove-redundant-endcf, on by default) -- correct as long as nothing runs in
between.  The register allocator, which runs later, may still place live-range-split COPIES in that block: they execute for no
lane, the registers keep the values of the split region, and whatever reads them afterwards reads garbage.  That is the wrong
model `kcf_update_sparse_run<7, true>` wrote (seven v_mov_b64 behind the loop of the second channel half's transform: the undo
of a register rotation; model slots j = 7, 9, 11, 13 of every thread wrong -- exactly the four registers of the rotation that
hold old model values).

The scan is mechanical.  Rule 1: every `s_cbranch_execnz` that jumps backwards and directly follows `s_andn2_b64 exec, exec, ...`
(a divergent loop's back edge) starts a region at its fall-through.  Rule 2: every forward `s_cbranch_execz` that directly follows
an instruction narrowing EXEC (`s_and_saveexec_b64`, `s_mov_b64 exec, ...`, `s_and_b64 exec, ...`: an `if`) starts a region at its
target, the join block.  A region ends at the next instruction that writes EXEC or branches; vector / memory instructions inside
it are reported (v_readlane / v_writelane / v_readfirstlane ignore EXEC and are not).  On this tree: 2,610 loop exits and
7,500 joins in the library's code objects, one function reported by both rules -- the one that computes the wrong model.

    python tools/isa_exec0_scan.py multiple-object-tracking_amd/libmot_amd.so        # the shipped code objects (llvm-objdump)
    python tools/isa_exec0_scan.py file.s ...                                        # hipcc -S --cuda-device-only output
Exit status 1 when a region was found.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
ARCH = os.environ.get("MOT_ARCH", "gfx950")          # the Makefile's ARCH; --arch overrides
_EXEC_WRITE = re.compile(r"^\s*s_\w+\s+exec(_lo|_hi)?\b|^\s*s_\w*saveexec\w*\s|^\s*s_\w*wrexec\w*\s")
_VECTOR = re.compile(r"^\s*(v_|ds_|flat_|global_|scratch_|buffer_|image_)\w+\s")
_NO_EXEC = re.compile(r"^\s*(v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)\s")
_BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+|s_endpgm|s_setpc_b64|s_swappc_b64)\b")
_SKIP = re.compile(r"^\s*(s_nop|s_waitcnt\w*)\b")
_LOOP_MASK = re.compile(r"^\s*s_andn2_b64\s+exec,\s*exec,")
_NARROW = re.compile(r"^\s*(s_and_saveexec_b64\s|s_mov_b64\s+exec,|s_and_b64\s+exec,)")


def _insn(line):
    """the instruction text of an assembly / disassembly line, or '' (labels, directives, comments, blank lines)"""
    s = line.split("//")[0].split(";")[0].rstrip()
    t = s.strip()
    if not t or t.startswith(".") or t.endswith(":") or re.match(r"^[0-9a-fA-F]+ <.*>:$", t): return ""
    return s


def scan_lines(lines, is_objdump):
    """([(rule, function, line number of the branch, [(line number, instruction), ...]), ...], sites examined)"""
    labels = {}; addr_line = {}; addr_of = {}
    for n, ln in enumerate(lines):
        if is_objdump:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            if m: a = int(m.group(1), 16); addr_line.setdefault(a, n); addr_of[n] = a
        else:
            m = re.match(r"^(\.L\w+):", ln)
            if m: labels[m.group(1)] = n

    def target(i, operand):
        """index of the line a branch at line i jumps to, or None"""
        if not is_objdump: return labels.get(operand)
        try: imm = int(operand, 0)
        except ValueError: return None
        if imm >= 0x8000: imm -= 0x10000                                # signed 16-bit word offset, printed unsigned
        return addr_line.get(addr_of.get(i, -(1 << 40)) + 4 + 4 * imm)

    def prev_insn(i):
        k = i - 1
        while k >= 0 and (not _insn(lines[k]) or _SKIP.match(_insn(lines[k]))): k -= 1
        return _insn(lines[k]) if k >= 0 else ""

    def region(first):
        """vector instructions from line `first` up to the next EXEC write or branch"""
        found = []
        for j in range(first, len(lines)):
            s = _insn(lines[j])
            if not s: continue
            if _EXEC_WRITE.match(s) or _BRANCH.match(s): break
            if _VECTOR.match(s) and not _NO_EXEC.match(s): found.append((j + 1, s.strip()))
        return found

    hits = []; func = None; sites = 0
    for i, ln in enumerate(lines):
        m = re.match(r"^(?:[0-9a-fA-F]+ <)?(_Z\w+)>?:", ln)
        if m: func = m.group(1)
        ins = _insn(ln)
        m = re.match(r"^\s*s_cbranch_exec(n?z)\s+(\S+)", ins)
        if not m: continue
        t = target(i, m.group(2))
        if t is None: continue
        if m.group(1) == "nz":
            # rule 1: the back edge of a divergent loop ("exec &= ~done; branch if any lane is left"): the fall-through has EXEC == 0
            if t < i and _LOOP_MASK.match(prev_insn(i)):
                sites += 1
                found = region(i + 1)
                if found: hits.append(("loop exit, EXEC == 0", func, i + 1, found))
        elif t > i and _NARROW.match(prev_insn(i)):
            # rule 2: "if" -- EXEC narrowed, skip when no lane is left: the join block must restore EXEC before any vector instruction
            # (it runs for the lanes of the "then" side only, or for none)
            sites += 1
            found = region(t)
            if found: hits.append(("join of an if, EXEC still narrowed", func, i + 1, found))
    return hits, sites


def code_objects(path):
    """the gfx950 code objects of a host library / object (clang offload bundles inside it), as byte strings"""
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and data[18:20] == struct.pack("<H", 224): return [data]          # an AMDGPU code object itself
    magic = b"__CLANG_OFFLOAD_BUNDLE__"; out = []; pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0: break
        num, = struct.unpack_from("<Q", data, i + 24); off = i + 32
        for _ in range(num):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + tl].decode(errors="replace"); off += tl
            if ARCH in triple and sz: out.append(data[i + o:i + o + sz])
        pos = i + len(magic)
    return out


def scan_file(path):
    if path.endswith(".s"):
        return scan_lines(open(path).read().split("\n"), False)
    hits = []; loops = 0
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        h, n = scan_lines(txt.split("\n"), True)
        hits += h; loops += n
    return hits, loops


def main(argv):
    """exit 1 on a hit, 2 when a file yields implausibly few sites (--min-sites N, default 1: a build gate that examines nothing -- another --offload-arch,
    a compressed offload bundle, a changed objdump listing -- must not pass vacuously; round-5 advisor finding)"""
    bad = 0; min_sites = 1; thin = 0
    if "--min-sites" in argv:
        i = argv.index("--min-sites"); min_sites = int(argv[i + 1]); argv = argv[:i] + argv[i + 2:]
    if "--arch" in argv:
        i = argv.index("--arch"); global ARCH; ARCH = argv[i + 1]; argv = argv[:i] + argv[i + 2:]
    for p in argv:
        hits, loops = scan_file(p)
        if loops < min_sites:
            thin += 1
            print(f"{p}: only {loops} sites examined (expected >= {min_sites}): no {ARCH} code object found, or the listing format changed")
        print(f"{p}: {loops} loop exits / if joins examined, {len(hits)} with vector instructions in front of the EXEC restore")
        for rule, func, line, found in hits:
            bad += 1
            print(f"  {func}  ({rule}; branch at line {line}): {len(found)} instruction(s) run for the wrong lanes")
            for n, s in found[:16]: print(f"      {n}: {s}")
    return 1 if bad else (2 if thin else 0)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
