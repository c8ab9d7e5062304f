"""Batch inference over the GPUs of one node: bucketed prompt batches (prompts.py) and the utterance-sharded sampler + gather (sharded.py)."""
from .prompts import get_inference_prompt, infer_prompts, padded_mel_batch, ragged_sample_fn, sample_kwargs, synthetic_metainfo  # noqa: F401
from .sharded import gather_utterances, sample_sharded, split_between_processes  # noqa: F401
