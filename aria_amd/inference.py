"""``python -m aria_amd.inference`` -- the reference's single-image inference script (aria/inference.py:30-153, README quick start) on the
MI355X path: same flags, same four steps (``load_model`` with an optional LoRA adapter, ``prepare_input`` through the chat template and the
processor, ``inference`` = sample until ``<|im_end|>``, decode).  Differences that follow from the path, not the interface: the checkpoint
is a local HF directory (no hub access), the adapter is the ``adapter_model.safetensors`` a ``use_peft`` run of ``aria_amd.train`` writes
and is folded into the base weights (``merge_and_unload``) so generation runs the fused prefill kernels and the native decode engine."""
from __future__ import annotations

import argparse


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="Aria inference on MI355X")
    parser.add_argument("--base_model_path", required=True, help="HF checkpoint directory (config.json + safetensors shards)")
    parser.add_argument("--peft_model_path", help="directory with adapter_config.json + adapter_model.safetensors (optional)")
    parser.add_argument("--tokenizer_path", required=True, help="tokenizer directory")
    parser.add_argument("--image_path", required=True)
    parser.add_argument("--prompt", required=True)
    parser.add_argument("--max_image_size", type=int, default=980)
    parser.add_argument("--split_image", action="store_true", default=False)
    parser.add_argument("--max_new_tokens", type=int, default=500)
    parser.add_argument("--temperature", type=float, default=0.9)
    return parser.parse_args(argv)


def load_model(base_model_path: str, peft_model_path: str = None, device="cuda"):
    from .modeling_aria import AriaForConditionalGeneration

    model = AriaForConditionalGeneration.from_pretrained(base_model_path, device=device)
    if peft_model_path:
        from .lora import load_lora_adapter, merge_and_unload

        load_lora_adapter(model, peft_model_path)
        merge_and_unload(model)
    return model.eval()


def prepare_input(image_path: str, prompt: str, processor, max_image_size: int, split_image: bool):
    from PIL import Image

    image = Image.open(image_path)
    messages = [{"role": "user", "content": [{"text": None, "type": "image"}, {"text": prompt, "type": "text"}]}]
    text = processor.apply_chat_template(messages, add_generation_prompt=True)
    return processor(text=text, images=image, return_tensors="pt", max_image_size=max_image_size, split_image=split_image)


def inference(image_path: str, prompt: str, model, processor, max_image_size: int = 980, split_image: bool = False, max_new_tokens: int = 500,
              temperature: float = 0.9) -> str:
    """aria/inference.py:101-131, call for call (the README quick start's generate arguments)."""
    inputs = prepare_input(image_path, prompt, processor, max_image_size, split_image)
    inputs["pixel_values"] = inputs["pixel_values"].to(model.dtype)
    inputs = {k: v.to(model.device) for k, v in inputs.items()}
    output = model.generate(**inputs, max_new_tokens=max_new_tokens, stop_strings=["<|im_end|>"], tokenizer=processor.tokenizer, do_sample=True,
                            temperature=temperature)
    prompt_len = inputs["input_ids"].shape[1]
    return processor.tokenizer.decode(output[0][prompt_len:].tolist(), skip_special_tokens=True).replace("<|im_end|>", "")


def main(argv=None):
    args = parse_arguments(argv)
    from .processing import AriaProcessor

    processor = AriaProcessor.from_pretrained(args.base_model_path, tokenizer_path=args.tokenizer_path)   # aria/inference.py:137-139
    model = load_model(args.base_model_path, args.peft_model_path)
    print(inference(args.image_path, args.prompt, model, processor, args.max_image_size, args.split_image, args.max_new_tokens, args.temperature))


if __name__ == "__main__":
    main()
