#!/usr/bin/env python3
"""Resolves the preprocessor conditionals of measured-and-not-kept experiments in the kernel sources: every macro in
UNDEF is taken as undefined (its #ifdef branch goes, its #ifndef / #else branch stays).  Run once per clean-up; the
variants stay in the repository's history (tools/sessions/*.sh name the commits and the measurements)."""
import re, sys
UNDEF = {
    "NSK_TAB_FLAT_LD", "NSK_TAB_FLAT_ST", "NSK_EP_WIN", "NSK_EP_W_NT", "NSK_LEARN_PREFETCH",
    "NSK_ABL_NOHUB", "NSK_ABL_NOATOMIC", "NSK_ABL_NODRAW", "NSK_ABL_LNOSINK", "NSK_ABL_LNOEV", "NSK_ABL_NOWALK",
    "NSK_ABL_NOAPPLY", "NSK_ABL_LNOEVST", "NSK_ABL_EPNOP1", "NSK_ABL_EPNOP2", "NSK_ABL_EPNOP3", "NSK_ABL_W_NOSTORE",
    "NSK_ABL_W_NOPHILOX", "NSK_ABL_W_NOLOAD", "NSK_ABL_NOPHILOX", "NSK_ABL_NOPASS2", "NSK_ABL_NOLUT", "NSK_ABL_LNOINIT",
    "NSK_ABL_LCHECK", "NSK_ABL_EPNOW", "NSK_ABL_EPNOVAL", "NSK_ABL_LNOBALLOT", "NSK_W_STORE_SC1", "NSK_W_STORE_NT",
}
KEEP_TIMING_IN = ("k_gibbs_seg_tabw",)      # NSK_ABL_TIMING stays only in the wide kernel (tools/timing_tabw.py)

def strip(path, drop_timing):
    lines = open(path).read().split("\n")
    out, stack = [], []          # stack entries: (kind, emitting_before, taken) kind: 'res' resolved / 'keep' untouched
    emitting = True
    fn = ""
    for ln in lines:
        m = re.match(r"\s*(?:template <.*>\s*)?(?:static )?__global__.*\bvoid (\w+)\(", ln)
        if m: fn = m.group(1)
        s = ln.strip()
        m1 = re.match(r"#\s*(ifdef|ifndef)\s+(\w+)", s)
        m2 = re.match(r"#\s*if\s+defined\((\w+)\)\s*(//.*)?$", s)
        m3 = re.match(r"#\s*elif\s+defined\((\w+)\)", s)
        if s.startswith("#if"):
            name, neg = None, False
            if m1: name, neg = m1.group(2), m1.group(1) == "ifndef"
            elif m2: name = m2.group(1)
            und = name in UNDEF or (name == "NSK_ABL_TIMING" and drop_timing and fn not in KEEP_TIMING_IN)
            if name and und:
                stack.append(("res", emitting, neg))
                emitting = emitting and neg       # ifdef X (undefined): skip; ifndef X: keep
                continue
            stack.append(("keep", emitting, None))
            if emitting: out.append(ln)
            continue
        if s.startswith("#elif") and stack and stack[-1][0] == "res":
            kind, before, taken = stack[-1]
            name = m3.group(1) if m3 else None
            assert name in UNDEF, (path, ln)
            continue                               # stays skipped until #else
        if s.startswith("#else") and stack and stack[-1][0] == "res":
            kind, before, taken = stack[-1]
            emitting = before and not taken
            stack[-1] = (kind, before, True)
            continue
        if s.startswith("#endif") and stack:
            kind, before, taken = stack.pop()
            if kind == "res":
                emitting = before
                continue
            if emitting: out.append(ln)
            continue
        if emitting: out.append(ln)
    assert not stack, path
    open(path, "w").write("\n".join(out))

for p in sys.argv[1:]:
    strip(p, True)
