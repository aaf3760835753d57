"""The reference's known-answer tests for ``apply_chat_template_and_tokenize`` (/root/reference/tests/test_apply_chat_template.py:18-222: the
ChatML string each conversation becomes, the image-token expansion per max_image_size, all-ignored labels for user-only samples, the error
for unknown content types) re-run against ``aria_amd.processing``.  The hub tokenizer of the reference tests is unreachable offline, so a
byte-level stub with the same special strings stands in; its decode is exact, so the expected strings are the reference's own."""
import pytest
import torch

from aria_amd import processing as P
from oracle.ref_processing import StubTokenizer

Q1, A1 = "Who wrote this book?\n", "Sylvie Covey"
Q2, A2 = "What is the title of this book?", "Modern Printmaking: A Guide to Traditional and Digital Techniques"


def user(*parts):
    return {"role": "user", "content": [{"text": None, "type": "image"} if p is None else {"text": p, "type": "text"} for p in parts]}


def assistant(text):
    return {"role": "assistant", "content": [{"text": text, "type": "text"}]}


def chatml(turns, n_img):
    """the expected string of the reference tests: <|im_start|>role\\n...<|im_end|>\\n per turn, an image = <fim_prefix> n x <|img|> <fim_suffix>"""
    out = ""
    for role, text in turns:
        out += f"<|im_start|>{role}\n" + text.replace("<image>", "<fim_prefix>" + "<|img|>" * n_img + "<fim_suffix>") + "<|im_end|>\n"
    return out


@pytest.fixture
def tok():
    t = StubTokenizer()
    t.pad_token = t.unk_token
    return t


def decode(tok, row, mask):
    ids = [int(i) for i, m in zip(row.tolist(), mask.tolist()) if m]
    out, buf = "", bytearray()
    for i in ids:
        if i < len(tok.SPECIAL):
            out += buf.decode("utf-8") + tok.SPECIAL[i]
            buf = bytearray()
        else:
            buf.append(i - 16)
    return out + buf.decode("utf-8")


ONE_ROUND = [user(Q1, None), assistant(A1)]
TWO_ROUNDS = ONE_ROUND + [user(Q2), assistant(A2)]
ONE_ROUND_S = [("user", Q1 + "<image>"), ("assistant", A1)]
TWO_ROUNDS_S = ONE_ROUND_S + [("user", Q2), ("assistant", A2)]


@pytest.mark.parametrize("size,n_img", [(980, 256), (490, 128)])
def test_single_user_message_expands_the_image_and_has_no_targets(tok, size, n_img):
    res = P.apply_chat_template_and_tokenize([[user(Q1, None)]], tok, iter([1]), max_image_size=size)
    assert decode(tok, res["input_ids"][0], res["attention_mask"][0]) == chatml([("user", Q1 + "<image>")], n_img)
    assert int((res["labels"] == -100).sum()) == res["input_ids"].numel()


def test_single_assistant_message(tok):
    res = P.apply_chat_template_and_tokenize([[assistant(A1)]], tok)
    assert decode(tok, res["input_ids"][0], res["attention_mask"][0]) == chatml([("assistant", A1)], 0)


@pytest.mark.parametrize("messages,turns", [(ONE_ROUND, ONE_ROUND_S), (TWO_ROUNDS, TWO_ROUNDS_S)])
def test_multi_turn_conversations(tok, messages, turns):
    res = P.apply_chat_template_and_tokenize([messages], tok, iter([1]))
    assert decode(tok, res["input_ids"][0], res["attention_mask"][0]) == chatml(turns, 256)
    supervised = decode(tok, res["input_ids"][0][res["labels"][0] != -100], torch.ones(int((res["labels"][0] != -100).sum())))
    assert supervised == "".join(text + "<|im_end|>\n" for role, text in turns if role == "assistant")  # aria/data.py:29-120: answers only


def test_invalid_content_type(tok):
    bad = {"role": "user", "content": [{"text": Q1, "type": "text"}, {"text": None, "type": "invalid"}]}
    with pytest.raises(ValueError) as err:
        P.apply_chat_template_and_tokenize([[bad]], tok)
    assert "Unknown content type invalid in message" in str(err.value)


def test_batch_of_conversations_is_padded_per_row(tok):
    res = P.apply_chat_template_and_tokenize([ONE_ROUND, TWO_ROUNDS], tok, iter([1, 1]))
    got = [decode(tok, r, m) for r, m in zip(res["input_ids"], res["attention_mask"])]
    assert got == [chatml(ONE_ROUND_S, 256), chatml(TWO_ROUNDS_S, 256)]
    assert res["input_ids"].shape[0] == 2 and int(res["attention_mask"][0].sum()) < int(res["attention_mask"][1].sum())
