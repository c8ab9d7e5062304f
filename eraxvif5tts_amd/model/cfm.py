"""Conditional-flow-matching sampler: drop-in for ``f5_tts.model.cfm.CFM`` on the inference path.

``sample()`` follows reference cfm.py:82-208 step by step on the host (mel of a raw-wave prompt, text -> ids, duration
rule, masks, per-sample seeded noise, sway-sampled time grid) and hands the whole ODE loop -- ``steps`` x (cond + uncond
DiT evaluation, CFG combine, Euler/midpoint update) and the final ``where(cond_mask, cond, y)`` -- to ONE native call
(``f5_sample`` in include/f5hip.h, hipGraph-replayed) when the backbone is the HIP ``DiT``.  For any other backbone
object (plug point A accepts arbitrary modules) the same loop is driven from Python over ``transformer(...)`` calls.
``forward()`` (the training loss, cfm.py:210-283) is outside the hot path and not provided.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils.rnn import pad_sequence

from .modules import MelSpec
from .utils import default, exists, lens_to_mask, list_str_to_idx, list_str_to_tensor


class CFM(nn.Module):
    def __init__(self, transformer, sigma=0.0, odeint_kwargs=dict(method="euler"), audio_drop_prob=0.35, cond_drop_prob=0.25,
                 num_channels=None, mel_spec_module=None, mel_spec_kwargs=dict(), frac_lengths_mask=(0.7, 1.0), vocab_char_map=None):
        super().__init__()
        self.frac_lengths_mask = frac_lengths_mask
        self.mel_spec = default(mel_spec_module, MelSpec(**mel_spec_kwargs))
        self.num_channels = default(num_channels, self.mel_spec.n_mel_channels)
        self.audio_drop_prob, self.cond_drop_prob = audio_drop_prob, cond_drop_prob
        self.transformer = transformer
        self.dim = transformer.dim
        self.sigma = sigma
        self.odeint_kwargs = odeint_kwargs
        self.vocab_char_map = vocab_char_map
        # where sample() draws its initial noise when none is given: None = on the model's device, as the reference does (cfm.py:182: a GPU
        # run uses the device generator); "cpu" = from torch's CPU generator and then moved, i.e. the numbers the reference's CPU path draws
        # for the same torch.manual_seed / seed= (end-to-end parity runs against the CPU oracle)
        self.noise_device = None

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, *args, **kwargs):
        raise NotImplementedError("CFM.forward is the training loss (reference cfm.py:210-283); only sample() is on the MI355X path")

    @torch.no_grad()
    def sample(self, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None, seed=None,
               max_duration=4096, vocoder=None, no_ref_audio=False, duplicate_test=False, t_inter=0.1, edit_mask=None,
               y0=None, return_trajectory=True, use_graph="auto", defer_guard=False, noise_device=None):
        """Same arguments and return value ``(out, trajectory)`` as the reference.  Extra keyword-only knobs:
        ``y0`` (explicit initial noise, zero-padded [b, N, mel]: parity tests), ``noise_device`` ("cpu": draw the noise as the reference's
        CPU path does, cfm.py:178-183 with self.device = cpu; default ``self.noise_device``, None = the model's device), ``return_trajectory=False`` skips
        materialising the [steps+1, b, N, mel] trajectory (returned as None), ``use_graph`` = True / False / "auto" (default: replay a hipGraph from the second call with the same shape on),
        ``defer_guard=True`` returns without the call's one stream synchronisation (several sample() calls can then be in flight on several
        streams); ``transformer.finish_pending()`` completes them."""
        self.eval()
        if cond.ndim == 2:  # raw wave
            cond = self.mel_spec(cond)
            cond = cond.permute(0, 2, 1)
            assert cond.shape[-1] == self.num_channels
        cond = cond.to(next(self.parameters()).dtype)
        batch, cond_seq_len, device = *cond.shape[:2], cond.device
        if not exists(lens):
            lens = torch.full((batch,), cond_seq_len, device=device, dtype=torch.long)

        if isinstance(text, list):
            if exists(self.vocab_char_map):
                text = list_str_to_idx(text, self.vocab_char_map).to(device)
            else:
                text = list_str_to_tensor(text).to(device)
            assert text.shape[0] == batch

        cond_mask = lens_to_mask(lens)
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask
        if isinstance(duration, int):
            duration = torch.full((batch,), duration, device=device, dtype=torch.long)
        duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration)  # at least one generated frame
        duration = duration.clamp(max=max_duration)
        max_dur = int(duration.amax())

        if duplicate_test:
            test_cond = F.pad(cond, (0, 0, cond_seq_len, max_dur - 2 * cond_seq_len), value=0.0)
        cond = F.pad(cond, (0, 0, 0, max_dur - cond_seq_len), value=0.0)
        if no_ref_audio:
            cond = torch.zeros_like(cond)
        cond_mask = F.pad(cond_mask, (0, max_dur - cond_mask.shape[-1]), value=False).unsqueeze(-1)
        step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
        mask = lens_to_mask(duration) if batch > 1 else None  # single inference needs no mask (cfm.py:152-155)

        if y0 is None:
            ndev = default(default(noise_device, self.noise_device), self.device)
            rows = []
            for dur in duration:
                if exists(seed):
                    torch.manual_seed(seed)
                rows.append(torch.randn(int(dur), self.num_channels, device=ndev, dtype=step_cond.dtype))
            y0 = pad_sequence(rows, padding_value=0, batch_first=True).to(device)
        else:
            y0 = y0.to(device=device, dtype=step_cond.dtype)

        t_start = 0
        if duplicate_test:
            t_start = t_inter
            y0 = (1 - t_start) * y0 + t_start * test_cond
            steps = int(steps * (1 - t_start))
        t = torch.linspace(t_start, 1, steps + 1, device=self.device, dtype=step_cond.dtype)
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
        method = self.odeint_kwargs.get("method", "euler")

        native = getattr(self.transformer, "native_sample", None)
        if native is not None and edit_mask is None:
            # lens-prefix cond_mask and duration-prefix key mask are rebuilt on the device by the kernels
            # a batch whose durations are all equal has an all-true key mask (cfm.py:152-155 builds it anyway): the kernels' unmasked
            # forms compute the same thing
            use_mask = mask is not None and bool((duration != max_dur).any())
            out, trajectory = native(cond, text, lens, duration, y0, t, steps, cfg_strength, method=method, use_mask=use_mask,
                                     return_trajectory=return_trajectory, use_graph=use_graph, defer_guard=defer_guard)
            out = out.to(step_cond.dtype)
        else:
            out, trajectory = self._sample_python(step_cond, cond, cond_mask, text, mask, y0, t, cfg_strength, method, return_trajectory)
        self.transformer.clear_cache()

        if exists(vocoder):
            out = out.permute(0, 2, 1)
            out = vocoder(out)
        return out, trajectory

    @torch.no_grad()
    def sample_ragged(self, cond, texts, durations, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None, seed=None,
                      max_duration=4096, y0s=None, noise_device=None, use_graph="auto"):
        """``sample()`` for several texts over ONE prompt, each with its own duration, in one set of kernel launches without padding the
        utterances to a common length (libf5hip ``f5_sample_ragged``).  Equivalent to ``[sample(cond, [t], d)[0] for t, d in zip(texts,
        durations)]`` -- the reference's batch-1 arithmetic per utterance (cfm.py:82-208 with batch = 1: no key mask), noise drawn in the
        same order -- and returns that list ([1, N_i, mel] each).  cond: mel [1, nc, mel] or raw wave [1, nw] (one prompt for every text), or
        [B, nc, mel] with ``lens`` (a prompt per utterance, zero-padded to a common nc as for ``sample()``); texts: list of str / list of token
        lists / id tensor; durations: list of ints."""
        self.eval()
        native = getattr(self.transformer, "native_sample_ragged", None)
        if native is None:
            raise NotImplementedError("the backbone has no ragged sampler")
        if cond.ndim == 2:  # raw wave
            cond = self.mel_spec(cond).permute(0, 2, 1)
        cond = cond.to(next(self.parameters()).dtype)
        nutt = len(texts)
        assert cond.shape[0] in (1, nutt) and cond.shape[-1] == self.num_channels
        cond_seq_len, device = cond.shape[1], cond.device
        if not exists(lens):
            lens = torch.full((nutt,), cond_seq_len, device=device, dtype=torch.long)
        if isinstance(texts, torch.Tensor):  # token ids [B, nt], -1 padded
            text = texts.to(device)
        elif len(texts) and isinstance(texts[0], torch.Tensor):  # one id row per utterance
            text = pad_sequence([x.reshape(-1) for x in texts], padding_value=-1, batch_first=True).to(device)
        elif exists(self.vocab_char_map):
            text = list_str_to_idx(list(texts), self.vocab_char_map).to(device)
        else:
            text = list_str_to_tensor(list(texts)).to(device)
        duration = torch.tensor([int(d) for d in durations], device=device, dtype=torch.long)
        duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration).clamp(max=max_duration)  # cfm.py:127-131
        frames = [int(d) for d in duration]
        conds, noises = [], []
        ndev = default(default(noise_device, self.noise_device), self.device)
        for i, n in enumerate(frames):
            conds.append(F.pad(cond[i if cond.shape[0] > 1 else 0], (0, 0, 0, n - cond_seq_len), value=0.0))
            if y0s is not None:
                noises.append(y0s[i].reshape(n, self.num_channels).to(device=device, dtype=cond.dtype))
            else:
                if exists(seed):
                    torch.manual_seed(seed)
                noises.append(torch.randn(n, self.num_channels, device=ndev, dtype=cond.dtype).to(device))
        t = torch.linspace(0, 1, steps + 1, device=self.device, dtype=cond.dtype)
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
        out = native(torch.cat(conds), text, lens, frames, torch.cat(noises), t, steps, cfg_strength, method=self.odeint_kwargs.get("method", "euler"),
                     use_graph=use_graph)
        self.transformer.clear_cache()
        return [o.unsqueeze(0).to(cond.dtype) for o in torch.split(out, frames)]

    def _sample_python(self, step_cond, cond, cond_mask, text, mask, y0, t, cfg_strength, method, return_trajectory):
        """Generic driver over ``transformer(...)`` calls (non-native backbones, edit_mask): same fixed-grid update rules."""
        def fn(tt, x):
            pred = self.transformer(x=x, cond=step_cond, text=text, time=tt, mask=mask, drop_audio_cond=False, drop_text=False, cache=True)
            if cfg_strength < 1e-5:
                return pred
            null_pred = self.transformer(x=x, cond=step_cond, text=text, time=tt, mask=mask, drop_audio_cond=True, drop_text=True, cache=True)
            return pred + (pred - null_pred) * cfg_strength

        y, traj = y0, [y0]
        for t0, t1 in zip(t[:-1], t[1:]):
            dt = t1 - t0
            if method == "euler":
                y = y + dt * fn(t0, y)
            elif method == "midpoint":
                half = 0.5 * dt
                y = y + dt * fn(t0 + half, y + fn(t0, y) * half)
            else:
                raise ValueError(f"unsupported fixed-grid ODE method: {method}")
            if return_trajectory:
                traj.append(y)
        out = torch.where(cond_mask, cond, y)
        return out, (torch.stack(traj) if return_trajectory else None)
