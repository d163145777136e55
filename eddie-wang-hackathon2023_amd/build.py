"""build.py: checkpoint -> engine directory.  Same CLI, same artefact names and the same JSON keys
as the reference's examples/whisper/build.py (W/build.py:26-31,42-143,145-397):

    <output_dir>/whisper_encoder_float16_tp1_rank0.engine     + encoder_config.json
    <output_dir>/whisper_decoder_<dtype>_tp1_rank0.engine     + decoder_config.json
    <output_dir>/whsiper_crossattn_float16_tp1_rank0.engine   + cross_attn_config.json   ([sic])
    <output_dir>/positional_embedding.npy

The `.engine` files hold our packed-weight blob (weight.py) in gfx950 layouts instead of a
TensorRT plan; the `*_config.json` files carry the keys the session wrappers read
(`builder_config.{precision,tensor_parallel,num_heads,hidden_size,num_layers,num_audio,
num_audio_ctx,num_text_ctx,vocab_size,use_int8_kv_cache,...}`, `plugin_config.
gpt_attention_plugin`; R/tensorrt_llm/builder.py:260-266).

The plugin flags (--use_gpt_attention_plugin / --use_gemm_plugin / --use_layernorm_plugin) are
accepted and persisted for CLI parity.  In the reference they do not change the Whisper graph
(they are set after tracing); here the HIP kernels are always the executed path.

Extra, not in the reference: `--synthetic {large-v2,tiny.en,micro,...}` builds from a seeded
random-init checkpoint (no checkpoint file exists on the boxes of this build), `--gelu tanh`
selects the TRT path's tanh GELU instead of the torch path's erf GELU.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import time
from collections import OrderedDict

import numpy as np
import torch

import weight as W
from weight import load_crossattn_linear_weight, load_decoder_weight, load_encoder_weight

logger = logging.getLogger("whisper_mi355")

MODEL_ENCODER_NAME = "whisper_encoder"
MODEL_DECODER_NAME = "whisper_decoder"
MODEL_CROSSATTN_NAME = "whsiper_crossattn"


def get_engine_name(model, dtype, tp_size, rank):
    return '{}_{}_tp{}_rank{}.engine'.format(model, dtype, tp_size, rank)


def serialize_engine(engine: bytes, path):
    logger.info(f'Serializing engine to {path}...')
    tik = time.time()
    with open(path, 'wb') as f:
        f.write(engine)
    logger.info(f'Engine serialized. Total time: {time.strftime("%H:%M:%S", time.gmtime(time.time() - tik))}')


def parse_arguments(args=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--world_size', type=int, default=1, help='world size, only 1 is supported (tensor_parallel=1)')
    parser.add_argument('--model_dir', type=str, default="large-v2.pt")
    parser.add_argument('--quantize_dir', type=str, default="quantize/1-gpu")
    parser.add_argument('--dtype', type=str, default='float16', choices=['float16', 'float32', 'bfloat16'])
    parser.add_argument('--log_level', type=str, default='info')
    parser.add_argument('--max_batch_size', type=int, default=256)
    parser.add_argument('--max_input_len', type=int, default=200)
    parser.add_argument('--max_output_len', type=int, default=200)
    parser.add_argument('--max_beam_width', type=int, default=1)
    parser.add_argument('--use_gpt_attention_plugin', nargs='?', const=None, type=str, default=False, choices=['float16'])
    parser.add_argument('--use_gemm_plugin', nargs='?', const=None, type=str, default=False,
                        choices=['float16', 'float32', 'bfloat16'])
    parser.add_argument('--use_layernorm_plugin', nargs='?', const=None, type=str, default=False,
                        choices=['float16', 'float32', 'bfloat16'])
    parser.add_argument('--output_dir', type=str, default='whisper_outputs')
    parser.add_argument('--use_weight_only', default=False, action="store_true",
                        help='Quantize weights for the various GEMMs to INT8.')
    parser.add_argument('--weight_only_precision', const='int8', type=str, nargs='?', default='int8',
                        choices=['int8', 'int4'])
    parser.add_argument('--int8_kv_cache', default=False, action="store_true",
                        help='By default, we use dtype for KV cache. int8_kv_cache chooses int8 quantization for KV')
    # extensions
    parser.add_argument('--int8_cross_kv', default=False, action="store_true",
                        help='BEYOND the reference (which keeps them fp16): store the cross-attention K/V as int8 codes, one scale '
                             'per layer from --quantize_dir (torch_whisper_convert.py -kv); halves the decode loop\'s HBM traffic')
    parser.add_argument('--synthetic', type=str, default=None, help='build from a seeded random-init checkpoint of this size')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--gelu', type=str, default='erf', choices=['erf', 'tanh'])
    args = parser.parse_args(args)
    logging.basicConfig(level=getattr(logging, str(args.log_level).upper(), logging.INFO))
    for plugin_arg in ['use_gemm_plugin', 'use_layernorm_plugin', 'use_gpt_attention_plugin']:
        if getattr(args, plugin_arg) is None:
            logger.info(f"plugin_arg is None, setting it as {args.dtype} automatically.")
            setattr(args, plugin_arg, args.dtype)
    if args.dtype != 'float16':
        raise ValueError("the gfx950 engine computes in float16 (the reference's Whisper build is float16 only)")
    if args.world_size != 1:
        raise ValueError("Whisper engines are built with tensor_parallel=1 (W/build.py:159,234)")
    return args


def _plugin_config(args) -> dict:
    """Shape of PluginConfig.__dict__ as saved by Builder.save_config (plugin/plugin.py:33-140)."""
    return OrderedDict(
        bert_attention_plugin=False, gpt_attention_plugin=args.use_gpt_attention_plugin,
        inflight_batching_gpt_attention_plugin=False, identity_plugin=False, gemm_plugin=args.use_gemm_plugin,
        smooth_quant_gemm_plugin=False, layernorm_plugin=args.use_layernorm_plugin,
        layernorm_quantization_plugin=False, rmsnorm_quantization_plugin=False, attention_qk_half_accumulation=False,
        remove_input_padding=False, context_fmha_type=0,
        weight_only_quant_matmul_plugin=(args.dtype if args.use_weight_only else False),
        nccl_plugin=False, quantize_per_token_plugin=False, quantize_tensor_plugin=False, paged_kv_cache=False,
        lookup_plugin=False, engine="whisper_mi355 (gfx950 HIP kernels)")


def save_config(builder_config: dict, plugin_config: dict, config_path: str):
    with open(config_path, 'w') as f:
        json.dump({'builder_config': builder_config, 'plugin_config': plugin_config}, f, indent=4)
    logger.info(f'Config saved to {config_path}.')


def _wo(args):
    """False, 'int8' or 'int4': what the weight loaders take (W/build.py:102-112 maps the same pair of flags to QuantMode)."""
    return args.weight_only_precision if args.use_weight_only else False


def _flags(args, int8_kv=False) -> int:
    f = W.FLAG_INT8_CROSS_KV if getattr(args, 'int8_cross_kv', False) else 0
    if args.use_weight_only:
        f |= W.FLAG_WEIGHT_ONLY_INT8
    if int8_kv:
        f |= W.FLAG_INT8_KV
    if args.gelu == 'tanh':
        f |= W.FLAG_GELU_TANH
    return f


def build_encoder(model, args):
    md, params = model['dims'], model['model_state_dict']
    builder_config = OrderedDict(
        name=MODEL_ENCODER_NAME, precision='float16', tensor_parallel=1, num_layers=md['n_audio_layer'],
        num_heads=md['n_audio_head'], hidden_size=md['n_audio_state'], max_batch_size=args.max_batch_size,
        int8=False, fp8=False, timing_cache=None, opt_level=None, use_refit=False, strongly_typed=False,
        num_mels=md['n_mels'], num_audio_ctx=md['n_audio_ctx'])
    tensors = load_encoder_weight(md, params, md['n_audio_layer'], use_weight_only=_wo(args))
    blob = W.serialize_engine_blob(W.ENGINE_ENCODER, _flags(args), md, tensors)
    save_config(builder_config, _plugin_config(args), os.path.join(args.output_dir, 'encoder_config.json'))
    serialize_engine(blob, os.path.join(args.output_dir, get_engine_name(MODEL_ENCODER_NAME, 'float16', 1, 0)))


def build_decoder(model, args):
    md, params = model['dims'], model['model_state_dict']
    positional_embedding = params['decoder.positional_embedding']
    if hasattr(positional_embedding, 'numpy'):
        positional_embedding = positional_embedding.detach().cpu().numpy()
    np.save(os.path.join(args.output_dir, 'positional_embedding.npy'), positional_embedding)
    builder_config = OrderedDict(
        name=MODEL_DECODER_NAME, precision=args.dtype, tensor_parallel=1, num_layers=md['n_text_layer'],
        num_heads=md['n_text_head'], num_audio=1, num_audio_ctx=md['n_audio_ctx'], num_text_ctx=md['n_text_ctx'],
        hidden_size=md['n_text_state'], vocab_size=md['n_vocab'], max_batch_size=args.max_batch_size,
        use_int8_kv_cache=bool(args.int8_kv_cache), use_int8_cross_kv=bool(args.int8_cross_kv), int8=bool(args.int8_kv_cache), fp8=False, timing_cache=None,
        opt_level=None, use_refit=False, strongly_typed=False)
    tensors = load_decoder_weight(params, md['n_text_layer'], args.quantize_dir,
                                  use_weight_only=_wo(args), use_int8_kv_cache=args.int8_kv_cache, use_int8_cross_kv=args.int8_cross_kv)
    blob = W.serialize_engine_blob(W.ENGINE_DECODER, _flags(args, args.int8_kv_cache), md, tensors)
    save_config(builder_config, _plugin_config(args), os.path.join(args.output_dir, 'decoder_config.json'))
    serialize_engine(blob, os.path.join(args.output_dir, get_engine_name(MODEL_DECODER_NAME, args.dtype, 1, 0)))


def build_crossattn_kv_linear(model, args):
    md, params = model['dims'], model['model_state_dict']
    builder_config = OrderedDict(
        name=MODEL_CROSSATTN_NAME, precision='float16', tensor_parallel=1, num_layers=md['n_text_layer'],
        num_heads=md['n_text_head'], int8=False, fp8=False, timing_cache=None, opt_level=None, use_refit=False,
        strongly_typed=False, hidden_size=md['n_text_state'], num_audio_ctx=md['n_audio_ctx'])
    tensors = load_crossattn_linear_weight(params, md['n_text_layer'], use_weight_only=_wo(args),
                                           use_int8_cross_kv=args.int8_cross_kv, quantize_dir=args.quantize_dir)
    blob = W.serialize_engine_blob(W.ENGINE_CROSS_KV, _flags(args), md, tensors)
    save_config(builder_config, _plugin_config(args), os.path.join(args.output_dir, 'cross_attn_config.json'))
    serialize_engine(blob, os.path.join(args.output_dir, get_engine_name(MODEL_CROSSATTN_NAME, 'float16', 1, 0)))


def build_from_checkpoint(model: dict, args):
    os.makedirs(args.output_dir, exist_ok=True)
    build_encoder(model, args)
    build_decoder(model, args)
    build_crossattn_kv_linear(model, args)


def run_build(args=None):
    args = parse_arguments(args)
    if args.synthetic:
        import synthetic
        model = synthetic.synthetic_checkpoint(args.synthetic, args.seed)
    else:
        model = torch.load(args.model_dir, map_location='cpu')
    build_from_checkpoint(model, args)


if __name__ == '__main__':
    run_build()
