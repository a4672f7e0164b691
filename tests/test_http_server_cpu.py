"""The HTTP surface (native /generate + OpenAI-compatible routes) on CPU, against a scripted engine:
request parsing, sampling-parameter defaults, batch and SSE streaming shapes, finish reasons and
usage accounting follow the reference (entrypoints/http_server.py:228-262, 491-515;
managers/tokenizer_manager.py:907-980; openai_api/adapter.py)."""
import json
import threading
import time

import pytest

pytest.importorskip("fastapi")
pytest.importorskip("httpx")
from fastapi.testclient import TestClient

from semi_pd_amd.entrypoints.http_server import build_app
from semi_pd_amd.managers.tokenizer_manager import TokenizerManager, sampling_params_from_dict
from semi_pd_amd.models.llama import LlamaConfig
from semi_pd_amd.server_args import ServerArgs

WORDS = ["<eos>", "alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "user:", "assistant:"]


class WordTokenizer:
    """Whitespace word-level tokenizer with the three methods the server uses."""
    eos_token_id = 0
    chat_template = None

    def encode(self, text):
        return [WORDS.index(w) for w in text.split()]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(WORDS[i] for i in ids if not (skip_special_tokens and i == 0))


class ScriptedEngine:
    """Emits token (last_prompt_token + step) % len(WORDS) one per poll; token 0 is EOS."""
    scheduler = None

    def __init__(self):
        self._outputs, self._finished, self._token_times, self._send_time = {}, {}, {}, {}
        self._logprobs = {}
        self.aborted = []
        self._live = {}
        self.requests = []
        self._lock = threading.Lock()

    def add_request(self, input_ids, sampling_params, rid=None, return_logprob=False, top_logprobs_num=0):
        with self._lock:
            self._outputs[rid], self._finished[rid], self._token_times[rid] = [], None, []
            if return_logprob:
                self._logprobs[rid] = {"token": [], "top": [], "k": top_logprobs_num}
            self._send_time[rid] = time.time()
            self._live[rid] = (list(input_ids), sampling_params)
            self.requests.append((rid, list(input_ids), sampling_params))
        return rid

    def poll(self, timeout=0.0):
        with self._lock:
            if not self._live:
                pass
            else:
                for rid in list(self._live):
                    ids, sp = self._live[rid]
                    if rid not in self._outputs:
                        del self._live[rid]
                        continue
                    tok = (ids[-1] + len(self._outputs[rid]) + 1) % len(WORDS)
                    self._outputs[rid].append(tok)
                    if rid in self._logprobs:  # scripted values: -0.25 * (position + 1); top-k = next ids
                        lp = self._logprobs[rid]
                        pos = len(lp["token"]) + 1
                        lp["token"].append(-0.25 * pos)
                        lp["top"].append([(-0.25 * pos - 0.5 * j, (tok + j) % len(WORDS)) for j in range(lp["k"])])
                    if tok == 0 and not sp.ignore_eos:
                        self._finished[rid] = "stop"
                    elif len(self._outputs[rid]) >= sp.max_new_tokens:
                        self._finished[rid] = "length"
                    if self._finished[rid] is not None:
                        del self._live[rid]
                time.sleep(0.001)  # one step of a real engine is not free; lets a front-end abort land mid-request
                return True
        time.sleep(min(timeout, 0.005))
        return False

    def abort_request(self, rid):
        with self._lock:
            self.aborted.append(rid)
            self._live.pop(rid, None)

    def check_children(self):
        pass

    def shutdown(self):
        pass


@pytest.fixture()
def client():
    cfg = LlamaConfig(vocab_size=len(WORDS), hidden_size=64, intermediate_size=64, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=2)
    sa = ServerArgs(model_config=cfg, context_length=64, served_model_name="word-model")
    eng = ScriptedEngine()
    tm = TokenizerManager(eng, sa, tokenizer=WordTokenizer())
    with TestClient(build_app(tm, sa)) as c:
        c.engine = eng
        yield c


def _sse_events(resp):
    out = []
    for line in resp.iter_lines():
        if line.startswith("data: "):
            out.append(line[6:])
    assert out[-1] == "[DONE]"
    return [json.loads(x) for x in out[:-1]]


def test_generate_text_and_ids(client):
    r = client.post("/generate", json={"text": "alpha beta", "sampling_params": {"max_new_tokens": 3}})
    assert r.status_code == 200
    body = r.json()
    # prompt ends with id 2 -> tokens 3, 4, 5
    assert body["output_ids"] == [3, 4, 5] and body["text"] == "gamma delta epsilon"
    meta = body["meta_info"]
    assert meta["finish_reason"] == {"type": "length", "length": 3}
    assert (meta["prompt_tokens"], meta["completion_tokens"], meta["cached_tokens"]) == (2, 3, 0)
    r = client.post("/generate", json={"input_ids": [1, 9], "sampling_params": {"max_new_tokens": 8}})
    body = r.json()  # 10, 11, 0 = EOS -> stop, EOS not rendered
    assert body["output_ids"] == [10, 11, 0] and body["text"] == "user: assistant:"
    assert body["meta_info"]["finish_reason"] == {"type": "stop", "matched": 0}
    # HTTP defaults of the reference: temperature 1.0, 128 new tokens (clipped to the context window)
    sp = client.engine.requests[-1][2]
    assert sp.temperature == 1.0 and sp.top_k == 1 << 30 and not sp.is_greedy


def test_generate_batch_and_stream(client):
    r = client.post("/generate", json={"text": ["alpha", "gamma delta"],
                                       "sampling_params": [{"max_new_tokens": 2}, {"max_new_tokens": 1, "temperature": 0}]})
    out = r.json()
    assert [o["output_ids"] for o in out] == [[2, 3], [5]]
    assert client.engine.requests[-1][2].is_greedy and not client.engine.requests[-2][2].is_greedy
    with client.stream("POST", "/generate", json={"text": "alpha", "stream": True,
                                                  "sampling_params": {"max_new_tokens": 4, "ignore_eos": True}}) as resp:
        ev = _sse_events(resp)
    assert ev[-1]["output_ids"] == [2, 3, 4, 5] and ev[-1]["meta_info"]["finish_reason"]["type"] == "length"
    assert all(e["meta_info"]["finish_reason"] is None for e in ev[:-1])
    lens = [len(e["output_ids"]) for e in ev]
    assert lens == sorted(lens) and ev[-1]["text"].startswith(ev[0]["text"])  # cumulative text


def test_generate_errors(client):
    assert client.post("/generate", json={"sampling_params": {}}).status_code == 400
    assert client.post("/generate", json={"input_ids": [99]}).status_code == 400            # outside the vocabulary
    assert client.post("/generate", json={"input_ids": list(range(1, 12)) * 7}).status_code == 400  # > context
    r = client.post("/generate", json={"text": "alpha", "sampling_params": {"top_p": 0.0}})
    assert r.status_code == 400 and "top_p" in r.json()["error"]["message"]
    r = client.post("/generate", json={"input_ids": [1], "sampling_params": {"repetition_penalty": 1.2}})
    assert r.status_code == 400
    with pytest.raises(ValueError):
        sampling_params_from_dict({"temperatur": 1})


def test_openai_completions(client):
    r = client.post("/v1/completions", json={"model": "word-model", "prompt": "alpha beta", "max_tokens": 2,
                                             "temperature": 0})
    body = r.json()
    assert body["object"] == "text_completion" and body["model"] == "word-model"
    assert body["choices"][0]["text"] == "gamma delta" and body["choices"][0]["finish_reason"] == "length"
    assert body["usage"] == {"prompt_tokens": 2, "completion_tokens": 2, "total_tokens": 4}
    # OpenAI default max_tokens = 16; batch of prompts; echo
    r = client.post("/v1/completions", json={"prompt": ["alpha", "eta"], "echo": True, "ignore_eos": True})
    body = r.json()
    assert len(body["choices"]) == 2 and body["usage"]["completion_tokens"] == 32
    assert body["choices"][0]["text"].startswith("alpha" + "beta gamma")  # echo = prompt text + completion text
    with client.stream("POST", "/v1/completions", json={"prompt": [1, 2], "max_tokens": 3, "stream": True}) as resp:
        ev = _sse_events(resp)
    assert "".join(e["choices"][0]["text"] for e in ev).split() == ["gamma", "delta", "epsilon"]
    assert ev[-1]["choices"][0]["finish_reason"] == "length" and ev[-1]["usage"]["total_tokens"] == 5


def test_openai_chat_and_models(client):
    r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "alpha beta"}],
                                                  "max_tokens": 2})
    body = r.json()
    # fallback template "user: alpha beta\nassistant:" -> last prompt token 11 -> tokens 0 (EOS): stop
    assert body["object"] == "chat.completion"
    assert body["choices"][0]["message"] == {"role": "assistant", "content": ""}
    assert body["choices"][0]["finish_reason"] == "stop" and body["usage"]["prompt_tokens"] == 4
    with client.stream("POST", "/v1/chat/completions",
                       json={"messages": [{"role": "user", "content": "alpha"}], "max_tokens": 3, "stream": True,
                             "ignore_eos": True}) as resp:
        ev = _sse_events(resp)
    assert ev[0]["choices"][0]["delta"]["role"] == "assistant"
    assert ev[-1]["choices"][0]["finish_reason"] == "length"
    assert client.get("/v1/models").json()["data"][0]["id"] == "word-model"
    assert client.get("/health").status_code == 200
    assert client.get("/health_generate").status_code == 200
    assert client.get("/get_model_info").json()["is_generation"] is True
    assert client.post("/flush_cache").status_code == 200


def test_cli_flags_of_the_reference_parse(tmp_path):
    """The reference's own launch line (evaluation/benchmark_deepseek_v2_lite_semi_pd.sh:14-16)."""
    import argparse
    from semi_pd_amd.server_args import add_cli_args, from_cli_args
    (tmp_path / "config.json").write_text(json.dumps({
        "architectures": ["LlamaForCausalLM"], "vocab_size": 320, "hidden_size": 64, "intermediate_size": 96,
        "num_hidden_layers": 2, "num_attention_heads": 4, "num_key_value_heads": 2, "max_position_embeddings": 256,
        "eos_token_id": 7, "torch_dtype": "bfloat16"}))
    argv = ["--model-path", str(tmp_path), "--trust-remote-code", "--context-length", "10240",
            "--watchdog-timeout", "60000", "--dist-timeout", "3600", "--enable-metrics", "--disable-radix-cache",
            "--served-model-name", "deepseek", "--mem-fraction-static", "0.82", "--tp", "1", "--enable-semi-pd"]
    sa = from_cli_args(add_cli_args(argparse.ArgumentParser()).parse_args(argv))
    assert sa.enable_semi_pd and sa.disable_radix_cache and sa.tp_size == 1 and sa.context_length == 10240
    assert sa.mem_fraction_static == 0.82 and sa.served_model_name == "deepseek" and sa.load_format == "auto"
    assert sa.model_config.num_key_value_heads == 2 and sa.eos_token_ids == [7]
    assert sa.triton_attention_num_kv_splits is None          # split count chosen per batch unless the flag is given
    sa = from_cli_args(add_cli_args(argparse.ArgumentParser()).parse_args(argv + ["--triton-attention-num-kv-splits", "16"]))
    assert sa.triton_attention_num_kv_splits == 16            # server_args.py:979-981
    # the CU policy of a CLI launch is the one the bench line measures (work-conserving shares, the reference's percentages)
    import bench
    assert (sa.cu_mask_mode, sa.prefill_backlog_full_tokens) == ("dynamic", bench.DEFAULT_BACKLOG_FULL_TOKENS)
    assert (sa.prefill_cu_percent, sa.decode_cu_percent) == (bench.DEFAULT_PREFILL_CU, bench.DEFAULT_DECODE_CU)
    sa = from_cli_args(add_cli_args(argparse.ArgumentParser()).parse_args(
        argv + ["--cu-mask-mode", "env", "--prefill-backlog-full-tokens", "0", "--prefill-cu-percent", "50",
                "--decode-cu-percent", "50", "--decode-stream-priority", "-1"]))
    assert (sa.cu_mask_mode, sa.prefill_backlog_full_tokens, sa.decode_stream_priority) == ("env", 0, -1)
    with pytest.raises(SystemExit):
        add_cli_args(argparse.ArgumentParser()).parse_args(argv + ["--cu-mask-mode", "static"])
    with pytest.raises(ValueError, match="cu_mask_mode"):
        ServerArgs(model_config=sa.model_config, cu_mask_mode="static")
    with pytest.raises(ValueError, match="stream priorities"):   # cu_share.py would drop the prioritised stream
        from_cli_args(add_cli_args(argparse.ArgumentParser()).parse_args(argv + ["--decode-stream-priority", "-1"]))


def test_logprobs_native_and_openai(client):
    r = client.post("/generate", json={"text": "alpha", "return_logprob": True, "top_logprobs_num": 2,
                                       "return_text_in_logprobs": True, "sampling_params": {"max_new_tokens": 3}})
    meta = r.json()["meta_info"]
    assert [x[:2] for x in meta["output_token_logprobs"]] == [[-0.25, 2], [-0.5, 3], [-0.75, 4]]
    assert meta["output_token_logprobs"][0][2] == "beta"
    assert [t[1] for t in meta["output_top_logprobs"][1]] == [3, 4] and meta["output_top_logprobs"][1][1][0] == -1.0
    # without top-k and without text
    r = client.post("/generate", json={"input_ids": [1], "return_logprob": True, "sampling_params": {"max_new_tokens": 2}})
    meta = r.json()["meta_info"]
    assert meta["output_token_logprobs"] == [[-0.25, 2, None], [-0.5, 3, None]] and meta["output_top_logprobs"] == [None, None]
    assert client.post("/generate", json={"text": "alpha", "return_logprob": True, "logprob_start_len": 0}).status_code == 400
    # OpenAI completions: logprobs = k
    r = client.post("/v1/completions", json={"prompt": "alpha", "max_tokens": 2, "logprobs": 1, "temperature": 0})
    lp = r.json()["choices"][0]["logprobs"]
    assert lp["tokens"] == ["beta", "gamma"] and lp["token_logprobs"] == [-0.25, -0.5]
    assert lp["top_logprobs"] == [{"beta": -0.25}, {"gamma": -0.5}] and lp["text_offset"] == [0, 4]
    # OpenAI chat
    r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "alpha"}], "max_tokens": 2,
                                                  "logprobs": True, "top_logprobs": 2, "ignore_eos": True})
    content = r.json()["choices"][0]["logprobs"]["content"]
    assert len(content) == 2 and content[0]["logprob"] == -0.25 and len(content[0]["top_logprobs"]) == 2
    assert client.post("/v1/completions", json={"prompt": "alpha", "max_tokens": 1}).json()["choices"][0]["logprobs"] is None


def test_stop_strings_cut_the_text_and_abort_the_request(client):
    r = client.post("/generate", json={"text": "alpha", "sampling_params": {"max_new_tokens": 400, "ignore_eos": True, "stop": ["delta", "zeta"]}})
    body = r.json()
    assert body["text"] == "beta gamma " and body["meta_info"]["finish_reason"] == {"type": "stop", "matched": "delta"}
    assert client.engine.aborted == [body["meta_info"]["id"]]          # the rest of the 400 tokens is not generated
    r = client.post("/v1/completions", json={"prompt": "alpha", "max_tokens": 400, "ignore_eos": True, "stop": "gamma"})
    c = r.json()["choices"][0]
    assert c["text"] == "beta " and c["finish_reason"] == "stop" and len(client.engine.aborted) == 2
    # a request that ends by itself is not aborted
    client.post("/generate", json={"text": "alpha", "sampling_params": {"max_new_tokens": 2, "stop": ["zeta"]}})
    assert len(client.engine.aborted) == 2


def test_parallel_sampling_n(client):
    """n > 1 (OpenAI `n`, native sampling_params.n): every prompt is run n times, choices are prompt-major, a prompt's
    tokens are counted once in `usage`."""
    r = client.post("/v1/completions", json={"prompt": "alpha", "max_tokens": 3, "n": 3})
    body = r.json()
    assert r.status_code == 200 and [c["index"] for c in body["choices"]] == [0, 1, 2]
    assert len({c["text"] for c in body["choices"]}) == 1            # the scripted engine is deterministic
    assert body["usage"]["prompt_tokens"] == 1 and body["usage"]["completion_tokens"] == 9
    r = client.post("/v1/completions", json={"prompt": ["alpha", "gamma delta"], "max_tokens": 2, "n": 2, "echo": True})
    texts = [c["text"] for c in r.json()["choices"]]
    assert len(texts) == 4 and texts[0] == texts[1] and texts[2] == texts[3]
    assert texts[0].startswith("alpha") and texts[2].startswith("gamma delta")
    assert r.json()["usage"]["prompt_tokens"] == 3
    r = client.post("/generate", json={"text": "alpha", "sampling_params": {"max_new_tokens": 2, "n": 2}})
    assert r.status_code == 200 and isinstance(r.json(), list) and len(r.json()) == 2
    r = client.post("/generate", json={"text": "alpha", "sampling_params": {"n": 0}})
    assert r.status_code == 400
