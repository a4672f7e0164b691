"""TokenizerManager: the asyncio front of the engine.  Tokenises a request, hands it to the decode
scheduler first and then to the prefill scheduler (the Semi-PD fan-out of
managers/tokenizer_manager.py:149-160, done inside Engine.add_request), collects the streamed token
ids and detokenises them incrementally (the reference runs the detokeniser as its own process,
managers/detokenizer_manager.py; here it shares the front-end process).

Output dictionaries follow tokenizer_manager.py:907-980:
  {"text": <cumulative text>, "meta_info": {"id", "finish_reason", "prompt_tokens",
   "completion_tokens", "cached_tokens", "e2e_latency"}}            (or "output_ids" when the server
runs with --skip-tokenizer-init)."""
from __future__ import annotations

import asyncio
import threading
import time
import uuid
from typing import Any, AsyncIterator, Dict, List, Optional, Union

from semi_pd_amd.managers.io_struct import SamplingParams

_SAMPLING_KEYS = ("max_new_tokens", "temperature", "top_p", "top_k", "min_p", "ignore_eos", "stop_token_ids",
                  "frequency_penalty", "presence_penalty", "min_new_tokens")


def get_tokenizer(path: str):
    """hf_transformers_utils.get_tokenizer: a local HF tokenizer directory (there is no network)."""
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, local_files_only=True)


def sampling_params_from_dict(d: Optional[dict]) -> SamplingParams:
    """sampling/sampling_params.py:33-58: defaults of the HTTP surface (temperature 1.0, 128 new tokens)."""
    d = dict(d or {})
    unknown = [k for k in d if k not in _SAMPLING_KEYS and k not in (
        "stop", "n", "skip_special_tokens", "spaces_between_special_tokens", "no_stop_trim",
        "repetition_penalty", "json_schema", "regex", "ebnf", "custom_params", "max_tokens")]
    if unknown:
        raise ValueError(f"unknown sampling parameters: {unknown}")
    if d.get("repetition_penalty", 1.0) != 1.0:
        # the reference accepts the field but has no penalizer for it (sampling/penaltylib/__init__.py): say so
        raise ValueError("repetition_penalty is not supported")
    if d.get("n", 1) != 1:
        raise ValueError("n > 1 reaches the sampler as n separate requests (generate_request expands it)")
    for k in ("json_schema", "regex", "ebnf"):
        if d.get(k):
            raise ValueError("structured output is not supported")
    kw = {k: d[k] for k in _SAMPLING_KEYS if k in d and d[k] is not None}
    kw.setdefault("temperature", 1.0)
    kw.setdefault("max_new_tokens", 128)
    return SamplingParams(**kw)


class _ReqState:
    def __init__(self, rid: str, prompt_tokens: int, loop: asyncio.AbstractEventLoop):
        self.rid = rid
        self.prompt_tokens = prompt_tokens
        self.event = asyncio.Event()
        self.loop = loop
        self.created = time.time()
        self.seen = 0


class TokenizerManager:
    def __init__(self, engine, server_args, tokenizer=None):
        self.engine = engine
        self.server_args = server_args
        self.tokenizer = tokenizer
        if tokenizer is None and not server_args.skip_tokenizer_init:
            self.tokenizer = get_tokenizer(server_args.tokenizer_path)
        self.states: Dict[str, _ReqState] = {}
        self._lock = threading.Lock()
        self._stop = False
        self._pump_error: Optional[BaseException] = None
        self.last_receive_tstamp = time.time()
        self._thread = threading.Thread(target=self._pump, name="semipd-output-pump", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------------------ output pump
    def _pump(self):
        eng = self.engine
        if getattr(eng, "model_runner", None) is not None:  # in-process (unified) engine steps here
            import torch
            torch.cuda.set_device(eng.model_runner.device)
        try:
            while not self._stop:
                progressed = eng.poll(timeout=0.02)
                if not progressed:
                    if eng.scheduler is not None:
                        time.sleep(0.001)
                    else:
                        eng.check_children()
                    continue
                self.last_receive_tstamp = time.time()
                with self._lock:
                    states = list(self.states.values())
                for st in states:
                    n = len(eng._outputs.get(st.rid, ()))
                    if n != st.seen or eng._finished.get(st.rid) is not None:
                        st.seen = n
                        st.loop.call_soon_threadsafe(st.event.set)
        except BaseException as e:  # surface scheduler death to every waiting request
            self._pump_error = e
            with self._lock:
                for st in self.states.values():
                    st.loop.call_soon_threadsafe(st.event.set)

    def shutdown(self):
        self._stop = True
        self._thread.join(timeout=5)

    # ------------------------------------------------------------------------------ requests
    def _tokenize(self, text: Optional[str], input_ids: Optional[List[int]]) -> List[int]:
        if input_ids is not None:
            return [int(t) for t in input_ids]
        if text is None:
            raise ValueError("either text or input_ids must be given")
        if self.tokenizer is None:
            raise ValueError("the server runs with --skip-tokenizer-init: send input_ids, not text")
        return self.tokenizer.encode(text)

    def _validate(self, ids: List[int], sp: SamplingParams):
        ctx = self.server_args.context_length
        if len(ids) == 0:
            raise ValueError("empty prompt")
        if len(ids) >= ctx:
            raise ValueError(f"The input ({len(ids)} tokens) is longer than the model's context length ({ctx} tokens).")
        vocab = self.server_args.model_config.vocab_size
        if max(ids) >= vocab or min(ids) < 0:
            raise ValueError("input_ids out of the vocabulary range")
        if sp.max_new_tokens is None or len(ids) + sp.max_new_tokens > ctx:
            sp.max_new_tokens = ctx - len(ids)  # tokenizer_manager.py: clipped to the context window
            if getattr(sp, "min_new_tokens", 0) > sp.max_new_tokens:
                sp.min_new_tokens = sp.max_new_tokens  # a clipped request must still be allowed to stop

    @staticmethod
    def _finish_dict(reason: Optional[str], completion_tokens: int, last_token: Optional[int]):
        if reason is None:
            return None
        if reason == "length":
            return {"type": "length", "length": completion_tokens}
        if reason == "stop":
            return {"type": "stop", "matched": last_token}
        return {"type": "abort", "message": str(reason)}

    def _token_text(self, tok: int) -> Optional[str]:
        return None if self.tokenizer is None else self.tokenizer.decode([tok])

    def _out_dict(self, st: _ReqState, ids: List[int], fin: Optional[str], skip_special_tokens: bool,
                  logprobs: Optional[dict] = None, text_in_logprobs: bool = False) -> dict:
        meta = {"id": st.rid, "finish_reason": self._finish_dict(fin, len(ids), ids[-1] if ids else None),
                "prompt_tokens": st.prompt_tokens, "completion_tokens": len(ids), "cached_tokens": 0}
        if logprobs is not None:
            # tokenizer_manager.py convert_logprob_style: (logprob, token id, token text or None) per token
            n = min(len(ids), len(logprobs["token"]))
            txt = self._token_text if text_in_logprobs else (lambda t: None)
            meta["output_token_logprobs"] = [(logprobs["token"][i], ids[i], txt(ids[i])) for i in range(n)]
            meta["output_top_logprobs"] = [[(lp, int(t), txt(int(t))) for lp, t in logprobs["top"][i]] or None
                                           for i in range(n)]
        if fin is not None:
            meta["e2e_latency"] = time.time() - st.created
        if self.tokenizer is None:
            return {"output_ids": list(ids), "meta_info": meta}
        shown = ids[:-1] if (fin == "stop" and ids) else ids  # the matched stop / EOS token is not rendered
        text = self.tokenizer.decode(shown, skip_special_tokens=skip_special_tokens)
        if fin is None and text.endswith("�"):
            text = text[:-1]  # an incomplete multi-byte character: wait for the next token
        return {"text": text, "output_ids": list(ids), "meta_info": meta}

    async def _one(self, text, input_ids, sampling: dict, stream: bool, rid: Optional[str],
                   return_logprob: bool = False, top_logprobs_num: int = 0,
                   text_in_logprobs: bool = False) -> AsyncIterator[dict]:
        sp = sampling_params_from_dict(sampling)
        stop = (sampling or {}).get("stop") or []
        stop = [stop] if isinstance(stop, str) else [s for s in stop if s]
        if stop and self.tokenizer is None:
            raise ValueError("stop strings need a tokenizer (server runs with --skip-tokenizer-init)")
        ids = self._tokenize(text, input_ids)
        self._validate(ids, sp)
        skip_special = bool((sampling or {}).get("skip_special_tokens", True))
        rid = rid or uuid.uuid4().hex
        st = _ReqState(rid, len(ids), asyncio.get_running_loop())
        with self._lock:
            if rid in self.states:
                raise ValueError(f"duplicate request id {rid}")
            self.states[rid] = st
            if return_logprob:
                self.engine.add_request(ids, sp, rid=rid, return_logprob=True, top_logprobs_num=int(top_logprobs_num))
            else:
                self.engine.add_request(ids, sp, rid=rid)
        completed = False
        try:
            sent = -1
            while True:
                await st.event.wait()
                st.event.clear()
                if self._pump_error is not None:
                    raise RuntimeError(f"scheduler failed: {self._pump_error}")
                out_ids = list(self.engine._outputs[rid])
                fin = self.engine._finished[rid]
                lps = getattr(self.engine, "_logprobs", {}).get(rid) if return_logprob else None
                if stop:
                    # schedule_batch.py:470-520 checks stop strings on the scheduler side with the
                    # tokenizer; here the front end owns the tokenizer, cuts the text and aborts the rest
                    out = self._out_dict(st, out_ids, fin, skip_special, lps, text_in_logprobs)
                    hit = min(((out["text"].find(s), s) for s in stop if s in out["text"]), default=None)
                    if hit is not None:
                        out["text"] = out["text"][: hit[0]]
                        out["meta_info"]["finish_reason"] = {"type": "stop", "matched": hit[1]}
                        out["meta_info"].setdefault("e2e_latency", time.time() - st.created)
                        completed = fin is not None
                        yield out
                        return
                if fin is not None:
                    completed = True
                    yield self._out_dict(st, out_ids, fin, skip_special, lps, text_in_logprobs)
                    return
                if stream and len(out_ids) != sent:
                    sent = len(out_ids)
                    yield self._out_dict(st, out_ids, None, skip_special, lps, text_in_logprobs)
        finally:
            if not completed and self.engine._finished.get(rid) is None:
                self.engine.abort_request(rid)  # client went away, or a stop string ended the request early
            with self._lock:
                self.states.pop(rid, None)
            for d in (self.engine._outputs, self.engine._finished, self.engine._token_times, self.engine._send_time,
                      getattr(self.engine, "_logprobs", {})):
                d.pop(rid, None)

    async def generate_once(self, obj: Dict[str, Any]) -> Union[dict, List[dict]]:
        """Non-streaming request: the single result, with the generator closed before returning."""
        g = self.generate_request(obj)
        try:
            return await g.__anext__()
        finally:
            await g.aclose()

    async def generate_request(self, obj: Dict[str, Any]) -> AsyncIterator[Union[dict, List[dict]]]:
        """obj follows GenerateReqInput (managers/io_struct.py:36-120): text | input_ids (single or batch),
        sampling_params (dict or list), stream, rid."""
        text, input_ids = obj.get("text"), obj.get("input_ids")
        stream = bool(obj.get("stream", False))
        sampling = obj.get("sampling_params") or {}
        rid = obj.get("rid")
        return_logprob = bool(obj.get("return_logprob", False))
        top_num = int(obj.get("top_logprobs_num") or 0)
        text_in_lp = bool(obj.get("return_text_in_logprobs", False))
        if return_logprob and obj.get("logprob_start_len", -1) not in (-1, None):
            raise ValueError("logprobs of prompt tokens (logprob_start_len >= 0) are not supported")
        if obj.get("token_ids_logprob"):
            raise ValueError("token_ids_logprob is not supported")
        is_batch = isinstance(text, list) or (isinstance(input_ids, list) and input_ids
                                               and isinstance(input_ids[0], list))
        # parallel sampling (sampling_params.n, io_struct.py:100-160 normalize_batch_and_arguments): every prompt is
        # sent n times, prompt-major; result i belongs to prompt i // n.  Without a radix cache the copies do not
        # share their prefill.
        n = int(sampling.get("n", 1) or 1) if isinstance(sampling, dict) else 1
        if n < 1:
            raise ValueError("n must be at least 1")
        if n > 1:
            sampling = dict(sampling, n=1)
            rep = lambda xs: [x for x in xs for _ in range(n)]  # noqa: E731
            if text is not None:
                text = rep(text if isinstance(text, list) else [text])
            else:
                input_ids = rep(input_ids if is_batch else [input_ids])
            if isinstance(rid, list):
                raise ValueError("explicit request ids cannot be combined with n > 1")
            rid, is_batch = None, True
        if not is_batch:
            g = self._one(text, input_ids, sampling, stream, rid, return_logprob, top_num, text_in_lp)
            try:
                async for out in g:
                    yield out
            finally:
                await g.aclose()  # run the per-request cleanup (and a pending abort) now, not at GC time
            return
        n = len(text) if isinstance(text, list) else len(input_ids)
        texts = text if isinstance(text, list) else [None] * n
        idss = input_ids if isinstance(input_ids, list) and input_ids and isinstance(input_ids[0], list) else [None] * n
        samplings = sampling if isinstance(sampling, list) else [sampling] * n
        rids = rid if isinstance(rid, list) else [None] * n
        if not (len(texts) == len(idss) == len(samplings) == len(rids) == n):
            raise ValueError("batch fields have different lengths")
        gens = [self._one(texts[i], idss[i], samplings[i], stream, rids[i], return_logprob, top_num, text_in_lp)
                for i in range(n)]
        if not stream:
            # one failing sub-request (e.g. a validation error) must not leave its siblings running: stop at the first
            # exception, cancel the rest, close every generator (per-request cleanup + abort of what is still in
            # flight), then re-raise
            tasks = [asyncio.ensure_future(g.__anext__()) for g in gens]
            try:
                done, pending = await asyncio.wait(tasks, return_when=asyncio.FIRST_EXCEPTION)
                for t in pending:
                    t.cancel()
                if pending:
                    await asyncio.gather(*pending, return_exceptions=True)
                for t in tasks:
                    if t in done and t.exception() is not None:
                        raise t.exception()
                results = [t.result() for t in tasks]
            finally:
                for t in tasks:
                    t.cancel()
                for g in gens:
                    try:
                        await g.aclose()
                    except RuntimeError:
                        pass  # generator still unwinding from the cancellation
            yield list(results)
            return
        queue: asyncio.Queue = asyncio.Queue()

        async def drain(i, g):
            try:
                async for out in g:
                    out["index"] = i
                    await queue.put(out)
            except Exception as e:  # noqa: BLE001
                await queue.put(e)
            await queue.put(None)

        tasks = [asyncio.create_task(drain(i, g)) for i, g in enumerate(gens)]
        done = 0
        try:
            while done < n:
                item = await queue.get()
                if item is None:
                    done += 1
                elif isinstance(item, Exception):
                    raise item
                else:
                    yield item
        finally:
            for t in tasks:
                t.cancel()
