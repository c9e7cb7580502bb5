"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's token-ids -> text step,
`Vocab::Vector2StringV2` (onnxruntime/src/vocab.cpp:164-305), `IsChinese` (:137-147, via Str2Int :116-125) and
`WordFormat` (:149-162), followed line by line.  Only tests may import it.

Pinning: vocab.cpp cannot be compiled in place (it includes <glog/logging.h> and <yaml-cpp/yaml.h>), and the
reference holds no fixture for it, so this restatement is pinned by the hand-derived known answers in
tests/test_vocab_text.py — each traced through the cited lines by hand.
"""
from __future__ import annotations

from typing import List


def is_chinese(ch: bytes) -> bool:
    # vocab.cpp:137-147: exactly three bytes, a well-formed 3-byte UTF-8 sequence, code point in [19968, 40959]
    if len(ch) != 3:
        return False
    if (ch[0] & 0xF0) != 0xE0 or (ch[1] & 0xC0) != 0x80 or (ch[2] & 0xC0) != 0x80:      # Str2Int returns 0 (:116-125)
        return False
    val = ((ch[0] & 0x0F) << 12) | ((ch[1] & 0x3F) << 6) | (ch[2] & 0x3F)
    return 19968 <= val <= 40959


def word_format(word: bytes) -> bytes:
    return {b"i": b"I", b"i'm": b"I'm", b"i've": b"I've", b"i'll": b"I'll"}.get(word, word)      # :149-162


class Vocab:
    """Byte-string vocabulary; keeps `last_is_complete_english_` between calls like the reference object (vocab.h:22)."""

    def __init__(self, tokens: List[str]):
        self.vocab = [t.encode("utf-8") for t in tokens]
        self.last_is_complete_english_ = False

    def vector2string_v2(self, ids: List[int], language: str = "") -> str:
        vocab = self.vocab
        words: List[bytes] = []
        is_pre_english = False
        pre_english_len = 0
        is_combining = False
        combine = b""
        unicode_char = "▁".encode("utf-8")
        first_word_need_space = self.last_is_complete_english_                    # :176
        n = len(ids)
        for i in range(n):
            word = vocab[ids[i]]
            if word in (b"<s>", b"</s>", b"<unk>"):                               # :181-182
                continue
            if language == "en-bpe":                                              # :183-198
                if word.find(unicode_char) != -1:
                    if combine != b"":
                        combine = word_format(combine)
                        if len(words) != 0:
                            combine = b" " + combine
                        words.append(combine)
                    combine = word[3:]
                else:
                    combine += word
                continue
            sub_word = word.find(b"@@") != -1                                     # :201
            if sub_word:
                if i < n - 1 and is_chinese(vocab[ids[i + 1]]):                   # :205-214
                    word = word[:len(word) - 2] + b" "
                    if is_combining:
                        combine += word
                        is_combining = False
                        word = combine
                        combine = b""
                elif i == n - 1:                                                  # :215-224
                    word = word[:len(word) - 2]
                    if is_combining:
                        combine += word
                        is_combining = False
                        word = combine
                        combine = b""
                    self.last_is_complete_english_ = False
                else:                                                             # :225-229
                    combine += word[:len(word) - 2]
                    is_combining = True
                    continue
            elif is_combining:                                                    # :232-237
                combine += word
                is_combining = False
                word = combine
                combine = b""
            if is_chinese(word):                                                  # :243-246
                words.append(word)
                is_pre_english = False
            else:
                if not is_pre_english and first_word_need_space:                  # :250-252
                    words.append(b" ")
                if not is_pre_english:                                            # :255-259
                    words.append(word)
                    pre_english_len = len(word)
                else:                                                             # :261-280
                    if pre_english_len > 1:
                        words.append(b" ")
                        words.append(word)
                        pre_english_len = len(word)
                    else:
                        if len(word) > 1:
                            words.append(b" ")
                        words.append(word)
                        pre_english_len = len(word)
                is_pre_english = True
            if i == n - 1 and not is_chinese(word) and not sub_word:              # :283-288
                self.last_is_complete_english_ = True
            else:
                self.last_is_complete_english_ = False
        if language == "en-bpe" and combine != b"":                               # :291-297
            combine = word_format(combine)
            if len(words) != 0:
                combine = b" " + combine
            words.append(combine)
        return b"".join(words).decode("utf-8", errors="surrogateescape")
