"""Host logic of bench.py that needs no GPU: the scalar side-config keys the driver's record keeps, the self-launching
`--gpus N`, and the tie between the committed PMC passes and the library that runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    sys.path.insert(0, ROOT)
    import bench as module

    module._TRAFFIC = None
    yield module
    module._TRAFFIC = None


def test_side_configs_become_scalar_config_keys(bench):
    e = {"ms_per_step": 6.9712345, "value": 2.1e7, "roofline": {"frac": 0.66123, "bound": "mfma", "step_frac": 0.65,
                                                                 "traffic": 84, "algorithmic_bytes": 12},
         "cpu_baseline": {"value": 31040.4}}
    flat = bench.side_scalars("cfg3a", e)
    assert flat == {"side_cfg3a_ms": 6.971, "side_cfg3a_frac": 0.661, "side_cfg3a_cpu_wps": 31000}
    assert all(isinstance(v, (int, float, str)) for v in flat.values())      # scalars only: dicts and lists are dropped
    assert bench.side_scalars("cfg5", {"workload": "cfg5", "error": "timed out after 420 s"}) == {"side_cfg5_error": "timed out after 420 s"}
    assert bench.side_summary_row(e) == [6.9712, 21000000, 0.6612, "mfma", 0.65, 7.0, 31040]
    # every configuration SURVEY 8(d) fixes is in the default side set (cfg4 r = 8, the EPS (3,6) colour model and the
    # headline model in the reference's own dtype included), and its scalars are among those that lead `config`
    want = {"cfg2_f32", "cfg1", "cfg3a", "cfg3b", "cfg4_r4", "cfg4_r8", "cfg4_r16", "cfg4_eps36", "cfg5"}
    assert set(bench.EXTRA_CONFIGS) >= want and set(bench.CONFIG_SCALARS) == want
    specs, image_size, q0, _ = bench.WORKLOADS["cfg4_eps36"]
    assert (specs, image_size, q0) == (((3, 6),), 32, 4) and bench.windows_per_sample(specs, image_size) * 128 == 115200
    assert bench.WORKLOADS["cfg2_f32"][0] == bench.WORKLOADS["cfg2"][0] and bench.WORKLOADS["cfg2_f32"][3] == __import__("torch").float32
    x = bench.synthetic_input(2, 32, 4, __import__("torch").float32, "cpu", 0)
    assert x.shape == (1, 2, 32, 32, 4) and bool((x[..., 3] == 1).all())   # the constant channel (dataset_loading.py:349-364)


def test_config_of_the_line_stays_within_what_the_driver_keeps(bench):
    """The driver's record kept ~900 characters of `config` (round 4: everything behind side_cfg1_frac was cut): the
    workload and three scalars for EVERY side configuration must fit 850 characters, prose lives elsewhere."""
    entries = [{"ms_per_step": 123.45678 / (i + 1), "value": 6.9e8 / (i + 1), "roofline": {"frac": 0.6543 / (i + 1), "bound": "mfma"},
                "cpu_baseline": {"value": 8412345.6 / (i + 1)}} for i in range(len(bench.EXTRA_CONFIGS))]
    base = {"workload": "cfg2 EPS[(3,4)]+linear bf16 B1024/GPU", "parallelism": "dp1"}
    cfg = bench.compact_config(base, bench.EXTRA_CONFIGS, entries)
    assert list(cfg)[:2] == ["workload", "parallelism"]
    for name in bench.CONFIG_SCALARS:
        assert {f"side_{name}_ms", f"side_{name}_frac", f"side_{name}_cpu_wps"} <= set(cfg)
    assert "side_cfg3a_bf16_ms" not in cfg
    assert len(json.dumps(cfg)) < 850, len(json.dumps(cfg))
    assert all(not isinstance(v, (dict, list)) and v is not None for v in cfg.values())


def test_gpus_n_without_world_size_starts_its_own_ranks(bench, monkeypatch, capsys):
    seen = {}

    class _Child:
        returncode = 0

        def communicate(self, timeout=None):
            return "RCCL banner\n" + json.dumps({"n_gpus": 4, "value": 1.0}) + "\n", None

    def fake_popen(cmd, cwd=None, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Child()

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as stop:
        bench.main()
    assert stop.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4            # ONE JSON line on stdout, nothing else
    import torch

    assert not torch.cuda.is_initialized()                                 # the parent made no GPU call


def test_pmc_traffic_is_null_unless_taken_on_the_loaded_library(bench, tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "library_sha256", lambda: "abc")
    table = {"_meta": {"so_sha256": "abc", "head": "deadbeef"}, "eps_bwd_dcore_q2reg_k:B1024": 123,
             "cfg5": [{"kernel": "lme_fold16_bwd_mfma_k<9>", "avg_us": 2.0, "traffic_bytes": 77}]}
    (prof / bench.TRAFFIC_FILE).write_text(json.dumps(table))
    assert bench.pmc_traffic("eps_bwd_dcore_q2reg_k:B1024") == 123 and bench.pmc_traffic("cfg5:lme_fold16_bwd") == 77
    assert bench.traffic_stamp() == ("deadbeef", True)
    r = bench.roofline_entry("hbm", "k", "c", 1e-3, 0, 1e6, None, traffic_key="eps_bwd_dcore_q2reg_k:B1024")
    assert r["traffic"] == 123 and r["traffic_head"] == "deadbeef" and r["traffic_on_this_library"] is True
    bench._TRAFFIC = None
    monkeypatch.setattr(bench, "library_sha256", lambda: "another build")
    assert bench.pmc_traffic("eps_bwd_dcore_q2reg_k:B1024") is None and bench.pmc_traffic("cfg5:lme_fold16_bwd") is None
    assert bench.traffic_stamp() == ("deadbeef", False)
