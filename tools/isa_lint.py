#!/usr/bin/env python3
"""Build-time check of the gfx950 assembly hipcc produced for a kernel source.

Two checks, both on compiler output rather than on source:

1. Lane-masked code ahead of an EXEC restore.  At a control-flow join the compiler restores the lanes that sat
   out the divergent region with `s_or_b64 exec, exec, s[..]` as the first thing in the join block.  Code the
   register allocator adds at the top of such a block (a copy that parks a caller-saved register around a call,
   a reload) belongs behind that instruction; placed in front of it, it runs for the lanes of the region only,
   and the others later read a register that was never written.  hipcc 7.2 (clang 22) does this now and then
   (sil_mask_kernel: `v_mov_b32 v43, v52` ahead of the restore in front of the call to sil_label; from a block's
   second view on, most lanes then addressed the histogram in LDS through a stale register).  Any instruction
   that depends on EXEC and sits between a block's label and its leading EXEC restore is reported, provided the
   label is the target of an `s_cbranch_execz` (the skip around the region, or a loop's exit: a real join).  A
   conditional block that the compiler laid out of line with a copy of the join's first instructions behind it
   looks the same but is only ever entered through `s_cbranch_execnz`, and is not reported.

2. Scratch budget: `--max-scratch KERNEL_SUBSTRING=BYTES` fails when a kernel's private segment grows beyond
   what the source was tuned for (a register-pressure regression that would otherwise only show as time).

Usage: isa_lint.py file.s [--max-scratch name=bytes ...]; exit status 1 on a finding.
"""
import re
import sys

EXEC_FREE = ("s_", "v_writelane_b32", "v_readlane_b32")   # do not depend on the lane mask
BRANCH = ("s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_endpgm", "s_call")
RESTORE = re.compile(r"^s_or_b64\s+exec,\s*exec,")
WRITES_EXEC = re.compile(r"^(s_\w+)\s+exec\b|^s_\w*saveexec|^v_cmpx")


def functions(lines):
    """(name, first line number, [(line number, text)]) per function body."""
    name, body, start = None, [], 0
    for no, raw in enumerate(lines, 1):
        text = raw.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", text)
        if m and not text.startswith(".L"):
            name, body, start = m.group(1), [], no
            continue
        if name and text.startswith(".Lfunc_end"):
            yield name, start, body
            name = None
            continue
        if name:
            body.append((no, text))


def lint_exec(path, lines):
    findings = []
    for name, _, body in functions(lines):
        joins = set()
        for _, text in body:
            m = re.match(r"\s*s_cbranch_execz\s+(\S+)", text)
            if m:
                joins.add(m.group(1))
        block, labelled = [], None
        def flush():
            if labelled not in joins:
                return
            lead = []
            for no, ins in block:
                if RESTORE.match(ins):
                    for bad_no, bad in lead:
                        findings.append(f"{path}:{bad_no}: {name}: `{bad}` runs ahead of the EXEC restore "
                                        f"`{ins}` (line {no}) of its block")
                    return
                if WRITES_EXEC.match(ins):
                    return
                if not ins.startswith(EXEC_FREE):
                    lead.append((no, ins))
        for no, text in body:
            ins = text.strip()
            if not ins or ins.startswith(".") and not ins.endswith(":"):
                continue
            if ins.endswith(":"):
                flush()
                block, labelled = [], ins[:-1]
                continue
            block.append((no, ins))
            if ins.startswith(BRANCH):
                flush()
                block, labelled = [], None
        flush()
    return findings


def lint_scratch(path, lines, limits):
    findings, kernel = [], None
    for raw in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", raw)
        if m:
            kernel = m.group(1)
        m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", raw)
        if m and kernel:
            for sub, limit in limits:
                if sub in kernel and int(m.group(1)) > limit:
                    findings.append(f"{path}: {kernel}: {m.group(1)} bytes of scratch per lane, budget {limit}")
    return findings


def main(argv):
    path, limits = None, []
    it = iter(argv[1:])
    for a in it:
        if a == "--max-scratch":
            sub, val = next(it).split("=")
            limits.append((sub, int(val)))
        else:
            path = a
    lines = open(path).read().split("\n")
    findings = lint_exec(path, lines) + lint_scratch(path, lines, limits)
    for f in findings:
        print("isa_lint:", f, file=sys.stderr)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
