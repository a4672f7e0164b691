"""The server's tokenizer path with a real HF fast tokenizer loaded from a local directory
(`get_tokenizer`, hf_transformers_utils.py of the reference): text in / text out through /generate,
the tokenizer's chat template in /v1/chat/completions, incremental detokenisation while streaming."""
import json

import pytest

pytest.importorskip("fastapi")
pytest.importorskip("httpx")
tokenizers = pytest.importorskip("tokenizers")
transformers = pytest.importorskip("transformers")
from fastapi.testclient import TestClient

from semi_pd_amd.entrypoints.http_server import build_app
from semi_pd_amd.managers.tokenizer_manager import TokenizerManager, get_tokenizer
from semi_pd_amd.models.llama import LlamaConfig
from semi_pd_amd.server_args import ServerArgs
from test_http_server_cpu import ScriptedEngine, WORDS, _sse_events


@pytest.fixture()
def tok_dir(tmp_path):
    vocab = {w: i for i, w in enumerate(WORDS)}
    vocab["<unk>"] = len(vocab)
    tk = tokenizers.Tokenizer(tokenizers.models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = tokenizers.pre_tokenizers.WhitespaceSplit()
    fast = transformers.PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<eos>", unk_token="<unk>")
    fast.chat_template = ("{% for m in messages %}{{ m['role'] }}: {{ m['content'] }} {% endfor %}"
                          "{% if add_generation_prompt %}assistant:{% endif %}")
    fast.save_pretrained(tmp_path)
    return str(tmp_path)


def test_server_with_hf_tokenizer_from_disk(tok_dir):
    tok = get_tokenizer(tok_dir)
    assert tok.encode("alpha beta") == [1, 2] and tok.eos_token_id == 0
    cfg = LlamaConfig(vocab_size=len(WORDS) + 1, hidden_size=64, intermediate_size=64, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=2)
    sa = ServerArgs(model_config=cfg, context_length=64, served_model_name="word-model", tokenizer_path=tok_dir)
    tm = TokenizerManager(ScriptedEngine(), sa)  # loads the tokenizer from tokenizer_path itself
    with TestClient(build_app(tm, sa)) as client:
        r = client.post("/generate", json={"text": "alpha beta", "sampling_params": {"max_new_tokens": 3}})
        body = r.json()
        assert body["output_ids"] == [3, 4, 5] and body["text"] == "gamma delta epsilon"
        # chat: the template renders "user: alpha beta assistant:" -> last id 11 -> EOS first: empty answer
        r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "alpha beta"}],
                                                      "max_tokens": 4})
        body = r.json()
        assert body["usage"]["prompt_tokens"] == 4 and body["choices"][0]["finish_reason"] == "stop"
        # streaming: deltas concatenate to the final text, special tokens are not rendered
        with client.stream("POST", "/v1/completions",
                           json={"prompt": "eta theta", "max_tokens": 6, "stream": True, "ignore_eos": True}) as resp:
            ev = _sse_events(resp)
        text = "".join(e["choices"][0]["text"] for e in ev)
        assert text.split() == ["iota", "user:", "assistant:", "alpha", "beta"]  # ids 9, 10, 11, 0 (<eos>, hidden), 1, 2
        assert ev[-1]["usage"]["completion_tokens"] == 6
