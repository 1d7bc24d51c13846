"""The reference-side host path: node -> compute_msm.js -> N-API shim -> C ABI -> HIP, on the golden
vectors, compared as decimal strings the way the harness does (src/ui/Benchmark.tsx:41-48).
GPU only; skipped when the image has no node."""
import json
import os
import shutil
import subprocess

import pytest

import pyref as R
import util

pytestmark = pytest.mark.gpu

NODE_DIR = os.path.join(util.ROOT, "webgpu-msm-bls12-377_amd", "node")


@pytest.mark.parametrize("name", ["g1_n33_random", "g1_n20_edge_scalars", "g1_n2_cancel", "g1_n1024_random"])
def test_compute_msm_js(golden, name):
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    addon = os.path.join(NODE_DIR, "build", "msm377_napi.node")
    assert os.path.exists(addon), "build the addon first: make -C webgpu-msm-bls12-377_amd/node"
    case = golden[name]
    proc = subprocess.run(
        [node, os.path.join(NODE_DIR, "run_golden.js"), os.path.join(util.GOLDEN_DIR, name + ".bin"), str(case["n"])],
        capture_output=True, text=True, timeout=300,
    )
    assert proc.returncode == 0, proc.stderr
    got = json.loads(proc.stdout.strip().splitlines()[-1])
    exp = R.decode_result(case["expected"])
    ex, ey = (0, 1) if exp is None else exp
    assert got["x"] == str(ex) and got["y"] == str(ey)
    assert got["empty_x"] == "0" and got["empty_y"] == "1"  # submission.ts:93-95
    assert got["version"].startswith("msm377")
    assert got["forms"] == 3  # Buffer, BigIntPoint[] / bigint[] and U32ArrayPoint[] / Uint32Array[] all went through the addon and agreed


@pytest.mark.parametrize("name", ["ed_n24_random", "ed_n10_edge_scalars", "ed_n2_cancel"])
def test_compute_msm_edwards_js(golden, name):
    """The Edwards-BLS12 twin through the addon (computeEdMsmSync -> msm377_ed_msm) on the Edwards golden cases."""
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    case = golden[name]
    proc = subprocess.run(
        [node, os.path.join(NODE_DIR, "run_golden.js"), os.path.join(util.GOLDEN_DIR, name + ".bin"), str(case["n"]), "ed"],
        capture_output=True, text=True, timeout=300,
    )
    assert proc.returncode == 0, proc.stderr
    got = json.loads(proc.stdout.strip().splitlines()[-1])
    assert got["x"] == str(int.from_bytes(case["expected"][:32], "little")) and got["y"] == str(int.from_bytes(case["expected"][32:], "little"))
    assert got["empty_x"] == "0" and got["empty_y"] == "1"
