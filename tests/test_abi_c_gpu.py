"""GPU: the C ABI driven from a plain C++ program (HIP runtime + libfdm_hip.so, no Python objects, no torch):
tests/abi_c/abi_smoke.cpp is compiled with hipcc against include/fdm_hip.h and run as a child process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_links_and_runs_against_the_library(tmp_path):
    from fdm_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    cmd = [hipcc, "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "abi_c", "abi_smoke.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lfdm_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout


def _write_records(path, recs):
    import struct
    import numpy as np
    with open(path, "wb") as f:
        for name, arr in recs.items():
            a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32)).reshape(-1)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<Q", a.size) + a.tobytes())


@pytest.mark.parametrize("dtype", [0, 2])          # FDM_F32, FDM_F16X3: the two modes inside the 1e-4 contract
def test_plan_layer_samples_a_clip_without_python(tmp_path, golden, dtype):
    """tests/abi_c/plan_smoke.cpp: model + case from a flat file -> fdm_plan_* / fdm_audio_prepare / fdm_sample_graph ->
    the reference's own chain outputs (tests/golden/chains_vocaset_tiny.npz) at 1e-4.  Python only writes the input file
    (seeded weights by reference state-dict name) and starts the process."""
    import numpy as np
    from fdm_amd import _lib
    from oracle import weights as W
    preset = "vocaset_tiny"
    g = golden(f"chains_{preset}")
    L = int(g["L"])
    inp = W.synth_inputs(preset, 1, L, seed=7)
    recs = {"w:" + k: v.numpy() for k, v in W.make_fdm_weights(preset).items()}
    recs.update(preset=np.frombuffer(preset.encode(), dtype=np.uint8), hub=inp["hub"].numpy(), style=inp["style"].numpy(),
                x_T=inp["x"].numpy(), noise=g["ddpm_lo_noise"], t_list=g["ddpm_lo_t"], expected_steps=g["ddpm_lo_steps"],
                expected_ddim3=g["ddim_3_final"], meta=[L, inp["hub"].shape[1], inp["hub"].shape[2], len(g["ddpm_lo_t"])])
    case = str(tmp_path / "case.bin")
    _write_records(case, recs)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "plan_smoke")
    cmd = [hipcc, "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "abi_c", "plan_smoke.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lfdm_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe, case, str(dtype)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plan_smoke ok" in r.stdout
    print(r.stdout.strip())


def test_two_rank_bench_on_one_gpu_matches_single_process(tmp_path):
    """The N > 1 path end to end without an 8-GPU node: `bench.py --gpus 2` as two fresh processes (gloo backend, both ranks
    on this box's one GPU, the launch contract's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment), against the
    single-process run of the same global batch: the gathered latents agree bit for bit and rank 0 prints one JSON line
    with n_gpus = 2.  (The RCCL branch itself needs a multi-GPU node: unmeasured until a SCALE record exists.)"""
    import json
    import socket
    import sys
    import numpy as np
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: child launches must start from a process that has not")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    bench = os.path.join(ROOT, "bench.py")
    common = [sys.executable, bench, "--config", "cfg1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--dtype", "f32"]
    one = str(tmp_path / "one.npy")
    r = subprocess.run(common + ["--gpus", "1", "--batch", "2", "--dump", one], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    two = str(tmp_path / "two.npy")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FDM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(common + ["--gpus", "2", "--batch", "1", "--dump", two, "--broadcast-weights"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 2
    assert rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo"        # self-checking: the line says how many ranks really ran
    assert rec["shard_check"] == {"rank": 1, "bit_identical": True}        # rank 0 recomputed rank 1's clip: the line verifies itself
    assert rec["weights"].startswith("broadcast")        # rank 1 started from another seed: bit-equal outputs below prove the broadcast
    assert rec["parity"]["max_abs"] < 1e-4 and rec["parity"]["dtype"] == "f32"
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (2, 1600, 64) and np.array_equal(a, b)
    # cfg5's gather: the end-to-end call (HuBERT-large -> tables -> a shortened chain -> quant -> decode) ends with the all-gather of
    # [clips, 498, 15069] fp32 vertices -- 60 MB per rank here, 120 MB at the benched 4 clips per GPU -- not the 3 MB of latents above
    common5 = [sys.executable, bench, "--config", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--headline-only", "--chain-steps", "6", "--dtype", "bf16"]
    one5, two5 = str(tmp_path / "one5.npy"), str(tmp_path / "two5.npy")
    r = subprocess.run(common5 + ["--gpus", "1", "--batch", "4", "--dump", one5], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        # (the two ranks co-run on this box's ONE GPU: this leg is what exposed the scalar-cache hazard of round 4's conv 0 kernel --
        #  two encoders sharing a device -- fixed in round 5)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FDM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(common5 + ["--gpus", "2", "--batch", "2", "--dump", two5], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    rec = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["config"]["global_batch"] == 4
    a, b = np.load(one5), np.load(two5)
    assert a.shape == b.shape == (4, 498, 15069) and np.isfinite(a).all() and np.array_equal(a, b)


def test_rccl_branch_of_bench_runs_with_one_rank(tmp_path):
    """The RCCL transport cannot be exercised across GPUs here (no multi-GPU node), but the branch that uses it can: `bench.py` with a
    ONE-rank `nccl` process group (FDM_DIST_FORCE=1) initialises RCCL on the device, gathers the clips with a device all-gather,
    reduces the timing with a device all-reduce and barriers through RCCL -- same line, same bits as the plain single-process run."""
    import json
    import socket
    import sys
    import numpy as np
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    bench = os.path.join(ROOT, "bench.py")
    common = [sys.executable, bench, "--config", "cfg1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--headline-only", "--dtype", "f16x3", "--gpus", "1"]
    one, two = str(tmp_path / "plain.npy"), str(tmp_path / "rccl.npy")
    r = subprocess.run(common + ["--dump", one], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FDM_DIST_FORCE="1")
    env.pop("FDM_DIST_BACKEND", None)
    r = subprocess.run(common + ["--dump", two], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["dist_backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["n_gpus"] == 1
    assert np.array_equal(np.load(one), np.load(two))


@pytest.mark.parametrize("config", ["shipped_biwi", "shipped_vocaset"])
def test_bench_shipped_configs_print_a_complete_line(config):
    """bench.py --config shipped_* (what the reference's samplers issue per test clip, end to end inside the timed call): one JSON
    line with the contract's keys, the stage split, and -- for the VOCASET style loop -- the sequential B = 1 pipelines beside the
    condition-batched call with bit-identical outputs."""
    import json
    import sys
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--config", config, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--headline-only",
                        "--dtype", "f16x3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline",
              "stages_ms"):
        assert k in d, k
    st = d["stages_ms"]
    assert set(st) >= {"audio_encoder", "tables", "quant", "decode", "sum"} and abs(st["sum"] - d["ms_per_step"]) / d["ms_per_step"] < 0.25
    assert config in d["config"]["workload"] and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    if config == "shipped_vocaset":
        assert d["sequential_loop"]["bit_identical_to_batched"] is True and d["speedup_vs_sequential_loop"] > 1.0
