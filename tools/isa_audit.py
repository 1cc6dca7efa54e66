#!/usr/bin/env python3
"""ISA audit of a gfx950 kernel (zero GPU minutes): compile lba_api.hip to assembly in the build container, cut one
kernel out, and bucket its instructions by class and - with the markers the SLSLAM_ISA_MARKERS build leaves in the
text - by phase of the tile loop.  Static counts; the per-phase execution weights (how often a phase runs per 64-lane
tile of the bench window) are applied by --weights.

  python tools/isa_audit.py --kernel k_linearise_schurILb0 [--markers] [--out profiles/round3_isa_audit_k1.txt]
"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "slslam_amd", "csrc", "lba_api.hip")

CLASSES = [
    ("fma_f64", re.compile(r"^v_fma_f64|^v_fmac_f64")),
    ("mul_f64", re.compile(r"^v_mul_f64")),
    ("add_f64", re.compile(r"^v_add_f64")),
    ("other_f64", re.compile(r"^v_(rsq|rcp|sqrt|min|max|cmp\w*|div_\w+|ldexp|frexp\w*|trig\w*|fract|floor|rndne|cvt\w*)_f64|^v_cmp\w*_f64|^v_cmpx\w*_f64|^v_cvt_\w*f64")),
    ("mfma", re.compile(r"^v_mfma")),
    ("dpp_mov", re.compile(r"^v_mov_b32_dpp|^v_mov_b32.*(row_shr|row_bcast|quad_perm|row_newbcast|row_mirror)")),
    ("mov", re.compile(r"^v_mov_b32|^v_mov_b64|^v_accvgpr|^v_pk_mov")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("int_valu", re.compile(r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|not|min|max|mbcnt|lshlrev|lshrrev|ashrrev|add3|lshl_add|lshl_or|and_or|or3|perm|alignbit|readlane|readfirstlane|writelane|cmp|cmpx)\w*_(u|i|b)(16|32|64)|^v_readlane|^v_readfirstlane|^v_writelane|^v_mbcnt|^v_cmp|^v_cmpx|^v_add_co|^v_addc_co|^v_lshl_add_u64|^v_mad_u64")),
    ("valu_other", re.compile(r"^v_")),
    ("ds_bpermute", re.compile(r"^ds_bpermute|^ds_permute")),
    ("ds_add_f64", re.compile(r"^ds_add_f64|^ds_add_rtn_f64|^ds_pk_add")),
    ("ds_read", re.compile(r"^ds_read|^ds_load")),
    ("ds_write", re.compile(r"^ds_write|^ds_store")),
    ("ds_other", re.compile(r"^ds_")),
    ("global_load", re.compile(r"^global_load|^buffer_load|^flat_load")),
    ("global_store", re.compile(r"^global_store|^buffer_store|^flat_store|^global_atomic")),
    ("scratch", re.compile(r"^scratch_")),
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("s_nop", re.compile(r"^s_nop")),
    ("salu", re.compile(r"^s_")),
]
ORDER = [c for c, _ in CLASSES]
VALU = ["fma_f64", "mul_f64", "add_f64", "other_f64", "mfma", "dpp_mov", "mov", "cndmask", "int_valu", "valu_other"]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return None


def compile_asm(markers, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only",
           "-S", "-o", out, SRC] + (["-DSLSLAM_ISA_MARKERS=1"] if markers else []) + extra
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def cut_kernel(path, key):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(key), l):
            start = i
        elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and re.match(r"^\s*s_endpgm", l) and end is None and False):
            pass
        if start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    assert start is not None and end is not None, "kernel %s not found" % key
    meta = {}
    for l in lines[end:end + 200]:
        m = re.match(r"\s*;\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs):\s*(\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    return lines[start:end], meta


def audit(body, loop_only):
    """returns {phase: Counter(class)}; phase markers are lines '; @PHASE name' (inline asm comments)."""
    phase = "prologue"
    out = collections.OrderedDict()
    for l in body:
        s = l.strip()
        m = re.match(r";\s*@PHASE\s+(\S+)", s)
        if m:
            phase = m.group(1)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        if c is None:
            c = "unclassified:" + op
        if c == "mov" and "dpp" in s:
            c = "dpp_mov"
        out.setdefault(phase, collections.Counter())[c] += 1
    return out


def fmt(table, weights, title):
    cols = ORDER
    seen = [c for c in cols if any(table[p].get(c) for p in table)]
    extra = sorted({c for p in table for c in table[p] if c not in cols})
    seen += extra
    lines = [title, ""]
    hdr = "%-22s %6s | %6s %6s" % ("phase", "weight", "VALU", "all") + " | " + " ".join("%11s" % c[:11] for c in seen)
    lines.append(hdr)
    lines.append("-" * len(hdr))
    tot = collections.Counter(); tot_valu = 0.0; tot_all = 0.0
    for p in table:
        w = weights.get(p, 1.0)
        valu = sum(table[p].get(c, 0) for c in VALU)
        allc = sum(table[p].values())
        lines.append("%-22s %6.2f | %6d %6d" % (p, w, valu, allc) + " | " + " ".join("%11d" % table[p].get(c, 0) for c in seen))
        for c in table[p]:
            tot[c] += w * table[p][c]
        tot_valu += w * valu; tot_all += w * allc
    lines.append("-" * len(hdr))
    lines.append("%-22s %6s | %6.0f %6.0f" % ("weighted / tile", "", tot_valu, tot_all) + " | " + " ".join("%11.0f" % tot.get(c, 0) for c in seen))
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_linearise_schurILb0")
    ap.add_argument("--markers", action="store_true", help="compile with -DSLSLAM_ISA_MARKERS=1 (phase markers + scheduling barriers)")
    ap.add_argument("--asm", default=None, help="use an existing .s instead of compiling")
    ap.add_argument("--weights", default="", help="phase=weight,... executions of the phase per tile (default 1; 0 drops prologue / epilogue)")
    ap.add_argument("--out", default=None)
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    path = a.asm or compile_asm(a.markers, a.extra)
    body, meta = cut_kernel(path, a.kernel)
    weights = {}
    for kv in filter(None, a.weights.split(",")):
        k, v = kv.split("=")
        weights[k] = float(v)
    table = audit(body, False)
    text = fmt(table, weights, "ISA audit of %s (%s build): static instruction counts per phase, weighted sum per 64-lane tile\n%s"
               % (a.kernel, "marker" if a.markers else "production", " ".join("%s=%d" % kv for kv in sorted(meta.items()))))
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
