#!/usr/bin/env python3
"""Instruction mix per kernel from a device ISA listing (hipcc -S --cuda-device-only).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/msm377.s webgpu-msm-bls12-377_amd/csrc/sequencer.hip
    python tools/isa_mix.py /tmp/msm377.s > profiles/r02_final/isa_mix.json

For every kernel: static counts of VALU instructions, v_mad_u64_u32, DPP moves, ds_bpermute, s_nop; and for the
accumulation kernels the same counts inside the innermost loop that holds a bucket addition (the loop with the most
v_mad_u64_u32), which is what bench.py's int32-mad roof multiplies by the number of additions."""
import json
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path, errors="ignore"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip("\n"))
    if name:
        yield name, body


def mix(lines):
    out = {"valu": 0, "v_mad_u64_u32": 0, "dpp": 0, "ds_bpermute": 0, "s_nop": 0, "global_load": 0, "global_store": 0}
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            out["valu"] += 1
        if op == "v_mad_u64_u32":
            out["v_mad_u64_u32"] += 1
        if "quad_perm" in t or "_dpp" in op:
            out["dpp"] += 1
        if op.startswith("ds_bpermute"):
            out["ds_bpermute"] += 1
        if op == "s_nop":
            out["s_nop"] += 1
        if op.startswith("global_load"):
            out["global_load"] += 1
        if op.startswith("global_store"):
            out["global_store"] += 1
    return out


def hottest_loop(lines):
    """Backward branches delimit loops: take the [label, branch] span with the most v_mad_u64_u32."""
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, ln in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = lines[labels[m.group(1)] : i + 1]
            mm = mix(span)
            if best is None or mm["v_mad_u64_u32"] > best["v_mad_u64_u32"]:
                best = mm
    return best


def demangle_hint(name):
    m = re.search(r"\d+(k_[a-z_0-9]+)", name)
    short = m.group(1) if m else name
    for tag in ("TeDev", "G1Dev", "EdDev", "TeAffBase", "AffWireSource", "AffDoublingSource"):
        if tag in name:
            short += "<%s>" % tag if "<" not in short else ""
    tags = [t for t in ("TeDev", "G1Dev", "EdDev", "TeAffBase", "AffWireSource", "AffDoublingSource") if t in name]
    return (m.group(1) if m else name) + ("<" + ",".join(tags) + ">" if tags else "")


def main():
    res = {}
    for name, body in kernels(sys.argv[1]):
        if "k_" not in name:
            continue
        entry = mix(body)
        if "k_accumulate" in name or "k_merge" in name or "k_reduce_tail" in name or "k_tree_step" in name:
            entry["hottest_loop"] = hottest_loop(body)
        key = demangle_hint(name)
        while key in res:
            key += "'"
        res[key] = entry
    json.dump(res, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
