"""bench.py's own code path (argument handling, timing hooks on the dominant kernel, barrier / max-over-ranks timing, roofline and JSON
assembly) exercised on CPU: tools/bench_dry_run.py hands bench.py a torch proxy whose "cuda" is the CPU, shrinks the model and routes
"nccl" to "gloo".  The numbers mean nothing; the line must carry every field of the driver's contract, for N = 1 and for two ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline"}


@pytest.mark.parametrize("world", [1, 2])
def test_bench_line_carries_the_contract(world):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_dry_run.py")] + (["--world", str(world)] if world > 1 else [])
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert CONTRACT <= set(res)
    assert res["n_gpus"] == world and res["steps"] == 2 and res["warmup"] == 1 and res["higher_is_better"] is True and res["scaling"] == "weak"
    assert res["value"] > 0 and res["ms_per_step"] > 0 and res["unit"] == "tokens/s" and res["dtype"] == "bf16"
    assert res["config"]["parallelism"] == ("single" if world == 1 else "dp2") and res["config"]["global_batch"] == 2 * world
    roof = res["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] == "mfma" and roof["launches_timed"] == 4
    assert roof["kernel"].startswith("gemm") and "experts.fc1" in roof["kernel"]
    if world == 1:  # the north_star target-shape sub-record rides in the same line (its code path, at toy size here)
        sub = res["long64k"]
        assert {"ms_per_step", "value", "unit", "steps", "roofline"} <= set(sub) and sub["steps"] == 3 and sub["value"] > 0
        assert sub["roofline"]["bound"] == "mfma" and sub["roofline"]["launches_timed"] == 3 * 2 and "INVALID" in sub
        assert res["recipe_grad_checkpointing"]["steps"] == 3 and res["recipe_grad_checkpointing"]["value"] > 0
        # BASELINE configs #2 / #4 ride in the same line too (gptfast surface of the same weights; toy lengths here)
        assert "inference_records_error" not in res and "sub_records_error" not in res, res
        gen, pre = res["generate_config2"], res["prefill_config4"]
        assert gen["value"] > 0 and gen["new_tokens"] == 6 and gen["runs"] == 1 and gen["warmup"] == 1 and gen["decode_engine"] is True   # (--sub-record-repeats 1: the protocol's 5 + 2 on hardware)
        assert gen["roofline"]["bound"] == "hbm" and gen["roofline"]["frac"] >= 0 and "INVALID" in gen
        assert pre["value"] > 0 and pre["finite_logits"] is True and pre["roofline"]["bound"] == "mfma" and "INVALID" in pre
        assert res["long64k"]["recompute_level"] == "moe" and res["recipe_grad_checkpointing"]["recompute_level"] == "moe"
        assert "recipe gradient checkpointing: OFF" in res["config"]["workload"]
        # the round-5 launch fusions off / on, alternating in this process (an A/B printed by bench.py itself)
        ab = res["step_fusions_ab"]
        assert "error" not in ab and ab["pairs"] == 1 and len(ab["off_runs_ms"]) == 1 == len(ab["on_runs_ms"]) and ab["on_ms_per_step"] > 0
        # recipes/config_lora.yaml on the same model: the adapters' step, the frozen-base floor, the recipe's checkpointing
        assert "lora_record_error" not in res, res.get("lora_record_error")
        lora = res["lora_config"]
        assert lora["ms_per_step"] > 0 and lora["frozen_base_fwd_dgrad_ms"] > 0 and lora["recipe_grad_checkpointing_ms"] > 0 and "INVALID" in lora
        assert lora["roofline"]["bound"] == "mfma" and "lora_dropout" not in lora and "dropout=0.05" in lora["workload"]
    else:
        assert "long64k" not in res and "generate_config2" not in res
