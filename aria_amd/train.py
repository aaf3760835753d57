"""Fine-tune entry point with the surface of the reference's ``aria/train.py`` (:46-253): ``python -m aria_amd.train --config
recipes/config_full.yaml`` (+ ``key=value`` / ``--key value`` overrides), launched one process per GPU
(``python -m torch.distributed.run --nproc-per-node N -m aria_amd.train ...``).

Honoured recipe keys (recipes/config_full.yaml): per_device_train_batch_size, gradient_accumulation_steps, learning_rate,
weight_decay, adam_beta2, warmup_ratio, lr_scheduler_type (cosine), max_seq_length, max_image_size, num_train_epochs / max_steps,
gradient_checkpointing, moe_z_loss_coeff, moe_aux_loss_coeff, freeze_vit, freeze_projector, freeze_llm, freeze_llm_layers, seed,
logging_steps, output_dir, dataset_mixer, save_strategy (epoch | steps | no) / save_steps, resume_from_checkpoint, use_peft + lora_*.  trl / peft / accelerate / DeepSpeed are replaced by: GradSync (RCCL all-reduce overlapped with backward),
ShardedAdamW (ZeRO-2-style sharded state, fused HIP AdamW), MoEAuxLossAutoScaler.set_loss_scale(1/grad_accum) (train.py:229).

Data: ``dataset_mixer`` (aria/data.py format and mixing rule: ``aria_amd.data``) through the reference's collate (chat template, label
masking, image processor: ``aria_amd.processing.collate_fn``), sharded by rank; ``synthetic_data=true`` (or no ``dataset_mixer``) draws
random samples of the configured shape instead (throughput runs, tests).  ``model_name_or_path`` may point to a local HF checkpoint
directory (``aria_amd.checkpoint``) or a ``torch.save``d state dict; the run ends with ``save_output`` (checkpoint directory or adapter).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver); before the HIP runtime starts

import torch  # noqa: E402


def load_config(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--synthetic", action="store_true", default=True)
    ap.add_argument("--tiny", action="store_true", help="toy dimensions (smoke tests)")
    args, rest = ap.parse_known_args(argv)
    cfg = dict(per_device_train_batch_size=8, gradient_accumulation_steps=2, learning_rate=5e-6, weight_decay=0.1, adam_beta2=0.95,
               warmup_ratio=0.01, lr_scheduler_type="cosine", max_seq_length=2048, max_image_size=980, num_train_epochs=1, max_steps=10,
               gradient_checkpointing=False, moe_z_loss_coeff=1e-5, moe_aux_loss_coeff=1e-3, freeze_vit=True, freeze_projector=False,
               freeze_llm=False, freeze_llm_layers=None, seed=42, logging_steps=1, output_dir="out", images_per_sample=2,
               max_grad_norm=1.0)  # HF TrainingArguments default, what zero2.yaml's gradient_clipping: auto resolves to
    if args.config:
        import yaml

        with open(args.config) as f:
            cfg.update({k: v for k, v in (yaml.safe_load(f) or {}).items()})
    it = iter(rest)
    for tok in it:
        if "=" in tok:
            k, v = tok.lstrip("-").split("=", 1)
        else:
            k, v = tok.lstrip("-"), next(it)
        try:
            v = json.loads(v)
        except Exception:
            pass
        cfg[k] = v
    cfg["tiny"] = args.tiny
    return cfg


def expert_parallel_groups(rank: int, world: int, ep: int):
    """DP x EP grid for ``expert_parallel_size = ep`` < world: ranks [d*ep, (d+1)*ep) form the expert-parallel group of data-parallel replica d
    (the all-to-all dispatch stays inside it -- on one node: neighbouring GPUs), ranks {r, r + ep, ...} hold the same expert shard.
    -> (this rank's expert-parallel group, its shard-replica group); (None, None) when ep == world (one group: the default).
    Every rank creates every group, in the same order (torch.distributed's rule)."""
    import torch.distributed as dist

    if ep <= 0 or world % ep:
        raise ValueError(f"expert_parallel_size {ep} does not divide the world size {world}")
    if ep == world:
        return None, None
    mine_ep = mine_dp = None
    for d in range(world // ep):
        g = dist.new_group(list(range(d * ep, (d + 1) * ep)))
        if rank // ep == d:
            mine_ep = g
    for r in range(ep):
        g = dist.new_group(list(range(r, world, ep)))
        if rank % ep == r:
            mine_dp = g
    return mine_ep, mine_dp


def build_model(cfg, device):
    from .modeling_aria import AriaConfig, AriaForConditionalGeneration
    from .moe_lm import AriaMoELMConfig
    from .vision import AriaVisionConfig

    path = cfg.get("model_name_or_path")
    from_dir = bool(path) and os.path.isdir(str(path)) and os.path.isfile(os.path.join(str(path), "config.json"))
    if from_dir:
        # HF checkpoint directory (config.json + sharded safetensors, reference key names): dimensions come from ITS config.json, like
        # AriaForConditionalGeneration.from_pretrained (aria/train.py:53-58), and every key must match -- a silently skipped tensor
        # would fine-tune from random weights
        model = AriaForConditionalGeneration.from_pretrained(str(path), device=device, strict=True)
        acfg = model.config
        t = acfg.text_config
        t.moe_z_loss_coeff, t.moe_aux_loss_coeff = cfg["moe_z_loss_coeff"], cfg["moe_aux_loss_coeff"]      # aria/train.py:67-68
        t.gradient_checkpointing = cfg["gradient_checkpointing"]
    else:
        if cfg["tiny"]:
            text = AriaMoELMConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=512, moe_intermediate_size=64,
                                   moe_num_experts=8, moe_topk=2, moe_z_loss_coeff=cfg["moe_z_loss_coeff"],
                                   moe_aux_loss_coeff=cfg["moe_aux_loss_coeff"], gradient_checkpointing=cfg["gradient_checkpointing"])
            side = int(cfg.get("tiny_image_size", 56))   # 56 -> 16 patches -> 4 tokens (synthetic); 490 -> 1225 -> 128 (what the processor emits)
            vis = AriaVisionConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, image_size=side)
            acfg = AriaConfig(vision_config=vis, text_config=text, projector_patch_to_query_dict={16: 4, 1225: 128, 4900: 256},
                              image_token_index=int(cfg.get("image_token_index", 9)))
        else:
            text = AriaMoELMConfig(moe_z_loss_coeff=cfg["moe_z_loss_coeff"], moe_aux_loss_coeff=cfg["moe_aux_loss_coeff"],
                                   gradient_checkpointing=cfg["gradient_checkpointing"])
            acfg = AriaConfig(vision_config=AriaVisionConfig(), text_config=text, image_token_index=int(cfg.get("image_token_index", 9)))
        torch.set_default_device(device)
        model = AriaForConditionalGeneration(acfg)
        torch.set_default_device("cpu")
        g = torch.Generator(device=device).manual_seed(cfg["seed"])
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "norm" in n and n.endswith("weight") or "ln_" in n and n.endswith("weight"):
                    p.fill_(1.0)
                elif n.endswith("bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, 0.02, generator=g)
        if path and os.path.isdir(str(path)):  # weight shards without a config.json: the configured dimensions, every key must match
            from .checkpoint import load_hf_dir_into

            load_hf_dir_into(model, str(path), strict=True)
        elif path and os.path.isfile(str(path)):  # a plain state-dict file
            sd = torch.load(path, map_location="cpu", weights_only=True)
            own = model.state_dict()
            missing = [k for k in own if k not in sd]
            if missing:
                raise KeyError(f"{path}: {len(missing)} parameters of the model are not in the file (first: {missing[:4]})")
            with torch.no_grad():
                for k, v in own.items():
                    if v.shape != sd[k].shape:
                        raise ValueError(f"{k}: file {tuple(sd[k].shape)} vs module {tuple(v.shape)}")
                    v.copy_(sd[k].to(v.dtype))
    if cfg["freeze_vit"]:
        model.freeze_vit()
    if cfg["freeze_projector"]:
        model.freeze_projector()
    if cfg["freeze_llm"]:
        model.freeze_llm()
    for i in (cfg.get("freeze_llm_layers") or []):
        for p in model.language_model.model.layers[int(i)].parameters():
            p.requires_grad = False
    if cfg.get("use_peft"):  # recipes/config_lora.yaml: adapters on the expert GEMMs and the LM's Linear projections (aria_amd/lora.py)
        from .lora import apply_lora_from_config

        skipped = apply_lora_from_config(model, cfg)
        if skipped and int(os.environ.get("RANK", "0")) == 0:
            print(f"[aria_amd.train] LoRA: {len(skipped)} target modules outside the language model left without adapter: {skipped[:4]}")
    return model.train(), acfg


def save_output(model, cfg) -> str:
    """``trainer.save_model(output_dir)`` (aria/train.py:247-249): full fine-tunes write the HF checkpoint directory (config.json + sharded
    safetensors, reference key names); ``use_peft`` runs write only the adapter, like peft's ``save_pretrained``: ``adapter_model.safetensors``
    (``...lora_A.weight`` / ``...lora_B.weight`` keyed by this package's module names) + ``adapter_config.json`` (r, alpha, dropout, targets)."""
    out = str(cfg["output_dir"])
    full = model.full_state_dict() if cfg.get("expert_parallel") else None   # a collective: every rank calls save_output in that case
    if int(os.environ.get("RANK", "0")) != 0:
        return out
    os.makedirs(out, exist_ok=True)
    if cfg.get("use_peft"):
        from safetensors.torch import save_file

        from .lora import lora_state_dict

        save_file({k: v.detach().cpu().contiguous() for k, v in lora_state_dict(model).items()}, os.path.join(out, "adapter_model.safetensors"),
                  metadata={"format": "pt"})
        with open(os.path.join(out, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LORA", "r": int(cfg.get("lora_r", 8)), "lora_alpha": int(cfg.get("lora_alpha", 32)),
                       "lora_dropout": float(cfg.get("lora_dropout", 0.0)), "target_modules": list(cfg.get("lora_target_modules") or []),
                       "base_model_name_or_path": cfg.get("model_name_or_path")}, f, indent=2)
    else:
        model.save_pretrained(out, state_dict=full)
    return out


def save_checkpoint(model, opt, cfg, step: int, history, rank: int, world: int) -> str:
    """``save_strategy`` checkpoints (recipes/config_full.yaml:20-21; HF Trainer layout ``output_dir/checkpoint-<step>``): the weights
    (rank 0; HF directory or adapter, as ``save_output``), every rank's optimizer shard, and ``trainer_state.json``."""
    path = os.path.join(str(cfg["output_dir"]), f"checkpoint-{step}")
    os.makedirs(path, exist_ok=True)
    save_output(model, {**cfg, "output_dir": path})          # writes on rank 0 (gathers expert shards from all ranks first if sharded)
    if rank == 0:
        with open(os.path.join(path, "trainer_state.json"), "w") as f:
            json.dump({"global_step": step, "world_size": world, "log_history": [float(x) for x in history]}, f)
    torch.save({**opt.state_dict(), "log_history": [float(x) for x in history]},   # (each rank logs the loss of ITS micro-batches)
               os.path.join(path, f"optimizer_rank{rank}.pt"))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()  # the checkpoint is complete (all shards written) before any rank moves on
    return path


def latest_checkpoint(output_dir: str):
    """The ``checkpoint-<step>`` directory with the highest step that is complete (has trainer_state.json), or None."""
    best = None
    if os.path.isdir(output_dir):
        for name in os.listdir(output_dir):
            if name.startswith("checkpoint-") and name[11:].isdigit() and os.path.exists(os.path.join(output_dir, name, "trainer_state.json")):
                if best is None or int(name[11:]) > int(best[11:]):
                    best = name
    return os.path.join(output_dir, best) if best else None


def load_checkpoint(model, opt, path: str, cfg, rank: int, world: int):
    """-> (step, history).  Weights first (they are what ShardedAdamW's bf16 side holds), then this rank's optimizer shard."""
    with open(os.path.join(path, "trainer_state.json")) as f:
        state = json.load(f)
    if state["world_size"] != world:
        raise ValueError(f"checkpoint written by {state['world_size']} ranks, resumed on {world}")
    if cfg.get("use_peft"):
        from safetensors.torch import load_file

        from .lora import lora_state_dict

        factors = load_file(os.path.join(path, "adapter_model.safetensors"))
        with torch.no_grad():
            for k, v in lora_state_dict(model).items():
                v.copy_(factors[k].to(v.dtype))
    else:
        from .checkpoint import load_checkpoint_dir, load_hf_into

        sd = load_checkpoint_dir(path)
        if cfg.get("expert_parallel"):  # the checkpoint holds all experts: keep this rank's
            for k in [k for k in sd if k.endswith(("mlp.experts.fc1.weight", "mlp.experts.fc2.weight"))]:
                ep = int(cfg.get("expert_parallel_size") or world)
                per, r = sd[k].shape[0] // ep, rank % ep
                sd[k] = sd[k][r * per:(r + 1) * per]
        load_hf_into(model, sd, strict=True)
    shard = torch.load(os.path.join(path, f"optimizer_rank{rank}.pt"), map_location="cpu")
    opt.load_state_dict(shard)
    return int(state["global_step"]), list(shard.get("log_history", state["log_history"]))


def synthetic_batch(cfg, acfg, device, gen):
    B, S = cfg["per_device_train_batch_size"], cfg["max_seq_length"]
    V = acfg.text_config.vocab_size
    R = acfg.vision_config.image_size if cfg["tiny"] else cfg["max_image_size"]
    P = (R // acfg.vision_config.patch_size) ** 2
    Q = acfg.projector_patch_to_query_dict[P]
    n_img = cfg["images_per_sample"]
    ids = torch.randint(10, V, (B, S), generator=gen, device=device)
    for j in range(n_img):
        ids[:, 4 + j * (Q + 2): 4 + j * (Q + 2) + Q] = acfg.image_token_index
    pv = torch.randn((B * n_img, 3, R, R), generator=gen, device=device).clamp_(-1, 1).to(torch.bfloat16)
    pm = torch.ones((B * n_img, R, R), dtype=torch.bool, device=device)
    labels = ids.clone()
    labels[:, : S // 2] = -100
    return dict(input_ids=ids, pixel_values=pv, pixel_mask=pm, attention_mask=torch.ones_like(ids), labels=labels)


def real_batches(cfg, acfg, device, rank, world, tokenizer=None, rows=None, skip: int = 0):
    """``dataset_mixer`` of the recipe -> device batches: aria/train.py:117-209 (collate: chat template + label masking + image processor)
    over the rows ``aria_amd.data.mix_datasets`` selects, sharded by rank."""
    import copy

    from .data import batches, mix_datasets
    from .processing import AriaVisionProcessor, collate_fn

    rows = mix_datasets(cfg["dataset_mixer"])["train"] if rows is None else rows

    class SizedProcessor(AriaVisionProcessor):
        """collate_fn calls ``processor(images, split_image=...)`` like aria/train.py:192, and ``__call__``'s own default (980,
        vision_processor.py:208) would override the configured size while the chat template expands ``processor.max_image_size``: with the
        recipe's 980 the two agree, with 490 the reference trips its token / feature count check.  Here the configured size is the default."""

        def __call__(self, images, max_image_size=None, **kw):
            return super().__call__(images, max_image_size=max_image_size, **kw)

    image_processor = SizedProcessor(max_image_size=int(cfg["max_image_size"]))
    max_steps = int(cfg.get("max_steps") or 0)
    epochs = 10 ** 9 if max_steps > 0 else float(cfg["num_train_epochs"])  # max_steps wins over num_train_epochs like in the HF Trainer
    for n, examples in enumerate(batches(rows, int(cfg["per_device_train_batch_size"]), rank, world, epochs)):
        if n < skip:  # resumed run: what the interrupted run consumed (the order is deterministic), skipped before any image is opened
            continue
        batch = collate_fn(copy.deepcopy(examples), tokenizer, image_processor, split_image=bool(cfg.get("split_image", False)),
                           max_seq_length=int(cfg["max_seq_length"]))
        yield {k: v.to(device) for k, v in batch.items() if k != "num_crops"}


def main(argv=None, tokenizer=None):
    cfg = load_config(sys.argv[1:] if argv is None else argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if use_cuda else "gloo")
    from .moe_lm import MoEAuxLossAutoScaler
    from .parallel import GradSync, ShardedAdamW, clip_scale, cosine_lr, global_grad_norm

    torch.manual_seed(int(cfg["seed"]))   # set_seed(training_args.seed): adapter init and dropout draw from the global generator -- the same on every rank
    model, acfg = build_model(cfg, device)
    ep_dp_group = None
    if cfg.get("expert_parallel") and world > 1:  # BASELINE config #5: routed experts sharded over an expert-parallel group, the rest data-parallel
        if cfg.get("use_peft"):
            raise NotImplementedError("expert_parallel with use_peft")
        ep_group, ep_dp_group = expert_parallel_groups(rank, world, int(cfg.get("expert_parallel_size") or world))
        model.enable_expert_parallel(ep_group)
    accum = int(cfg["gradient_accumulation_steps"])
    aux_scale_before = MoEAuxLossAutoScaler.main_loss_backward_scale
    MoEAuxLossAutoScaler.set_loss_scale(1.0 / accum)                     # aria/train.py:229 (a process-wide setting, restored on return)
    # ZeRO-2 (recipes/accelerate_configs/zero2.yaml): gradients reduce-scattered onto the rank that owns their optimizer shard
    sync = GradSync(model, mode=str(cfg.get("grad_exchange", "reduce_scatter")), ep_dp_group=ep_dp_group) if world > 1 else None
    opt = ShardedAdamW(model.named_parameters(), lr=cfg["learning_rate"], betas=(0.9, cfg["adam_beta2"]), weight_decay=cfg["weight_decay"])
    gen = torch.Generator(device=device).manual_seed(cfg["seed"] + rank)
    # data: the recipe's dataset_mixer (aria/data.py format) unless synthetic_data=true / no dataset is configured (throughput runs, tests)
    use_real = bool(cfg.get("dataset_mixer")) and not cfg.get("synthetic_data", False)
    total = int(cfg.get("max_steps") or 0)
    rows, steps_per_epoch = None, None
    if use_real:
        from .data import mix_datasets

        if tokenizer is None:
            from transformers import AutoTokenizer

            tokenizer = AutoTokenizer.from_pretrained(str(cfg.get("tokenizer_path") or cfg["model_name_or_path"]), use_fast=False)
        if getattr(tokenizer, "pad_token", None) is None:
            tokenizer.pad_token = tokenizer.unk_token        # aria/train.py:226-227

        rows = mix_datasets(cfg["dataset_mixer"])["train"]
        steps_per_epoch = max(1, len(rows) // world // int(cfg["per_device_train_batch_size"]) // accum)
        if total <= 0:  # epochs: optimizer steps = batches per rank // accumulation
            total = int(float(cfg["num_train_epochs"]) * (len(rows) // world // int(cfg["per_device_train_batch_size"]))) // accum
    history, first = [], 1
    strategy = str(cfg.get("save_strategy", "no"))
    save_every = steps_per_epoch if strategy == "epoch" else int(cfg.get("save_steps", 500)) if strategy == "steps" else None
    if cfg.get("resume_from_checkpoint"):  # true: the latest checkpoint in output_dir (config_full.yaml:21); a path: that one
        ck = cfg["resume_from_checkpoint"]
        ck = latest_checkpoint(str(cfg["output_dir"])) if ck is True else str(ck)
        if ck:
            done, history = load_checkpoint(model, opt, ck, cfg, rank, world)
            first = done + 1
            if not use_real and not cfg.get("synthetic_fixed"):
                for _ in range(done * accum):  # replay the generator draws of the steps already taken
                    synthetic_batch(cfg, acfg, device, gen)
    stream = real_batches(cfg, acfg, device, rank, world, tokenizer, rows, skip=(first - 1) * accum) if use_real else None
    for step in range(first, total + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss_acc = 0.0
        for micro in range(accum):
            if use_real:
                batch = next(stream)
            else:
                if cfg.get("synthetic_fixed"):  # overfit one batch (tests): random labels are irreducible otherwise
                    gen.manual_seed(cfg["seed"] + rank + micro)
                batch = synthetic_batch(cfg, acfg, device, gen)
            # only the last micro-step of the accumulation window exchanges gradients (DeepSpeed's gradient-accumulation boundary)
            with (sync.no_sync() if sync is not None and micro < accum - 1 else contextlib.nullcontext()):
                out = model(**batch, return_logits=False, validate_image_tokens=use_real)
                (out.loss / accum).backward()
            loss_acc += float(out.loss.detach()) / accum
        if sync is not None:
            sync.finish()
        # gradient clipping as the recipe runs it (zero2.yaml:5 gradient_clipping: auto -> max_grad_norm 1.0): one norm over the owned
        # slices of every rank, applied as the optimizer's gradient scale (nothing rewrites the 49.8 GB of gradients)
        gnorm = global_grad_norm(opt.params, sync) if cfg.get("max_grad_norm") else 0.0
        opt.step(lr=cosine_lr(step, total, cfg["learning_rate"], cfg["warmup_ratio"]), grad_scale=clip_scale(gnorm, cfg.get("max_grad_norm")))
        if use_cuda:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        history.append(loss_acc)
        if save_every and step % save_every == 0 and step < total:
            save_checkpoint(model, opt, cfg, step, history, rank, world)
        if rank == 0 and step % int(cfg["logging_steps"]) == 0:
            toks = world * accum * cfg["per_device_train_batch_size"] * cfg["max_seq_length"]
            print(json.dumps({"step": step, "loss": round(loss_acc, 4), "lr": opt.lr, "grad_norm": round(gnorm, 4), "step_s": round(dt, 3), "tokens_per_s": round(toks / dt, 1)}),
                  flush=True)
    if cfg.get("save_final", not cfg["tiny"]):  # written by rank 0: every rank holds the full updated bf16 weights (ShardedAdamW all-gathers)
        save_output(model, cfg)
        if rank == 0 and use_real:  # processor.save_pretrained(output_dir), aria/train.py:247: image-processor config + tokenizer files
            from .processing import AriaProcessor, AriaVisionProcessor

            AriaProcessor(image_processor=AriaVisionProcessor(max_image_size=int(cfg["max_image_size"])), tokenizer=tokenizer).save_pretrained(
                str(cfg["output_dir"]))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    MoEAuxLossAutoScaler.set_loss_scale(aux_scale_before)
    return history


if __name__ == "__main__":
    main()
