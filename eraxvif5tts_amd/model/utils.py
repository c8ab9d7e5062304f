"""Text front-end and tensor helpers of the inference path (host side, bit-identical id tensors).

Mirrors ``f5_tts/model/utils.py`` of the reference: ``lens_to_mask`` (:42-47), ``list_str_to_tensor`` (:81-84),
``list_str_to_idx`` (:88-95), ``get_tokenizer`` (:118-241), ``convert_char_to_pinyin`` (:243-284), ``seed_everything`` (:18-25).
"""
from __future__ import annotations

import os
import random
import re

import torch
from torch.nn.utils.rnn import pad_sequence


def seed_everything(seed=0):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def lens_to_mask(t, length=None):
    """bool [b, n]: position < length of each row (reference utils.py:42-47)."""
    if length is None:
        length = t.amax()
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def list_str_to_tensor(text, padding_value=-1):
    """utf-8 byte tokenizer (reference utils.py:81-84)."""
    rows = [torch.tensor([*bytes(t, "UTF-8")]) for t in text]
    return pad_sequence(rows, padding_value=padding_value, batch_first=True)


def list_str_to_idx(text, vocab_char_map, padding_value=-1):
    """char / pinyin tokenizer: unknown symbols map to index 0, batch padding is -1 (reference utils.py:88-95)."""
    rows = [torch.tensor([vocab_char_map.get(c, 0) for c in t]) for t in text]
    return pad_sequence(rows, padding_value=padding_value, batch_first=True)


def get_tokenizer(path_or_dataset_name, tokenizer_type="pinyin"):
    """Returns (vocab_char_map, vocab_size) exactly as the reference does for its vocab.txt format:
    one token per line, index = order of first occurrence, line 0 kept verbatim when it is a single space,
    every other line stripped, duplicates keep their first index (reference utils.py:118-241)."""
    if tokenizer_type == "custom":
        if os.path.isfile(path_or_dataset_name):
            vocab_path = path_or_dataset_name
        else:
            cand = os.path.join(path_or_dataset_name, "vocab.txt")
            if os.path.isdir(path_or_dataset_name) and os.path.isfile(cand):
                vocab_path = cand
            else:
                raise FileNotFoundError(
                    "Custom tokenizer type specified, but the provided path is not a valid file or directory "
                    f"containing vocab.txt: '{path_or_dataset_name}'")
    elif tokenizer_type in ("pinyin", "char"):
        base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "..", "data")
        cands = [os.path.join(base, f"{path_or_dataset_name}_{tokenizer_type}", "vocab.txt"),
                 os.path.join(base, path_or_dataset_name, "vocab.txt"),
                 os.path.join(base, f"Emilia_ZH_EN_{tokenizer_type}", "vocab.txt")]
        vocab_path = next((c for c in cands if os.path.isfile(c)), None)
        if vocab_path is None:
            raise FileNotFoundError(f"Default vocab file not found for dataset '{path_or_dataset_name}' and type '{tokenizer_type}'")
    elif tokenizer_type == "byte":
        return None, 256
    else:
        raise ValueError(f"Unknown tokenizer type: {tokenizer_type}")

    vocab_char_map = {}
    with open(vocab_path, "r", encoding="utf-8") as f:
        for i, line in enumerate(f):
            raw = line.rstrip("\n\r")
            tok = raw if (i == 0 and raw == " ") else raw.strip()
            if tok not in vocab_char_map:
                vocab_char_map[tok] = len(vocab_char_map)
    if not vocab_char_map:
        raise ValueError(f"Vocabulary file '{vocab_path}' resulted in zero processed tokens.")
    return vocab_char_map, len(vocab_char_map)


# ----------------------------------------------------------------------------- convert_char_to_pinyin
_TRANS = str.maketrans({";": ",", "“": '"', "”": '"', "‘": "'", "’": "'"})
_RE_BLOCK = re.compile(r"([一-鿕a-zA-Z0-9+#&\._%\-]+)")
_RE_SKIP = re.compile(r"(\r\n|\s)")
_RE_ENG = re.compile(r"([a-zA-Z0-9]+(?:\.\d+)?%?)")
_RE_HAN = re.compile(r"[一-鿕]")


def _segment_no_han(text):
    """Word segmentation for text WITHOUT Han characters, restating what jieba.cut() does on such input
    (jieba is a third-party dependency absent from this image; its dictionary only matters for Han text):
    blocks of [A-Za-z0-9+#&._%-] are split into alphanumeric runs and the symbol runs between them, whitespace is
    emitted as single tokens and every other character (accented letters, punctuation) on its own."""
    out = []
    for blk in _RE_BLOCK.split(text):
        if not blk:
            continue
        if _RE_BLOCK.fullmatch(blk):
            if len(blk) == 1:
                out.append(blk)
            else:
                out.extend(x for x in _RE_ENG.split(blk) if x)
        else:
            for x in _RE_SKIP.split(blk):
                if not x:
                    continue
                if _RE_SKIP.fullmatch(x):
                    out.append(x)
                else:
                    out.extend(x)
    return out


def _is_chinese(c):
    return "㄀" <= c <= "鿿"


def convert_char_to_pinyin(text_list, polyphone=True):
    """Reference utils.py:243-284.  ASCII / Vietnamese / other alphabetic text passes through character by character
    (with the reference's space insertion before multi-letter ASCII runs); Han text needs jieba + pypinyin, which are
    used when importable and otherwise raise (there is no silent approximation of pinyin)."""
    try:
        import jieba  # type: ignore
        from pypinyin import Style, lazy_pinyin  # type: ignore
        if jieba.dt.initialized is False:
            jieba.default_logger.setLevel(50)
            jieba.initialize()
        have_zh = True
    except Exception:  # noqa: BLE001
        jieba = None
        have_zh = False

    final = []
    for text in text_list:
        chars = []
        text = text.translate(_TRANS)
        if have_zh:
            segs = jieba.cut(text)
        else:
            if _RE_HAN.search(text):
                raise RuntimeError("Han characters need the jieba and pypinyin packages (not installed); "
                                   "ASCII / Vietnamese text does not")
            segs = _segment_no_han(text)
        for seg in segs:
            nbytes = len(bytes(seg, "UTF-8"))
            if nbytes == len(seg):  # pure alphabets and symbols
                if chars and nbytes > 1 and chars[-1] not in " :'\"":
                    chars.append(" ")
                chars.extend(seg)
            elif polyphone and nbytes == 3 * len(seg) and have_zh:  # pure east asian characters
                py = lazy_pinyin(seg, style=Style.TONE3, tone_sandhi=True)
                for i, c in enumerate(seg):
                    if _is_chinese(c):
                        chars.append(" ")
                    chars.append(py[i])
            else:  # mixed characters
                for c in seg:
                    if ord(c) < 256:
                        chars.extend(c)
                    elif _is_chinese(c):
                        if not have_zh:
                            raise RuntimeError("Han characters need the jieba and pypinyin packages")
                        chars.append(" ")
                        chars.extend(lazy_pinyin(c, style=Style.TONE3, tone_sandhi=True))
                    else:
                        chars.append(c)
        final.append(chars)
    return final
