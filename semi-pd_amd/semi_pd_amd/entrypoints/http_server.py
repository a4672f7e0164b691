"""HTTP front end: the native /generate API and the OpenAI-compatible endpoints, same routes and JSON
shapes as the reference (entrypoints/http_server.py:145-160 health, :199-218 model/server info,
:228-262 /generate incl. SSE streaming, :289-297 /flush_cache, :491-515 /v1/completions,
/v1/chat/completions, /v1/models; openai_api/adapter.py for the OpenAI payloads).

    python -m semi_pd_amd.launch_server --model-path <dir> --enable-semi-pd [--load-format dummy]

The server process hosts the TokenizerManager; the schedulers are the Engine's processes (prefill and
decode instance per TP rank in Semi-PD mode)."""
from __future__ import annotations

import asyncio
import json
import logging
import time
import uuid
from typing import Any, AsyncIterator, Dict, List, Optional

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse

logger = logging.getLogger(__name__)


def _error(message: str, status: int = 400, err_type: str = "BadRequestError") -> JSONResponse:
    """http_server.py:596-599 / adapter.py create_error_response."""
    return JSONResponse({"error": {"message": message, "type": err_type, "code": status}}, status_code=status)


def _sse(obj) -> bytes:
    return b"data: " + json.dumps(obj, ensure_ascii=False).encode() + b"\n\n"


def _openai_finish(meta: dict) -> Optional[str]:
    fr = meta.get("finish_reason")
    return None if fr is None else fr.get("type", "stop")


def _openai_sampling(body: dict, default_max_tokens: Optional[int]) -> dict:
    """adapter.py v1_generate_request / v1_chat_generate_request: OpenAI fields -> sampling_params."""
    sp = {"temperature": body.get("temperature", 1.0), "top_p": body.get("top_p", 1.0)}
    mt = body.get("max_completion_tokens", body.get("max_tokens", default_max_tokens))
    if mt is not None:
        sp["max_new_tokens"] = int(mt)
    for k in ("top_k", "min_p", "ignore_eos", "stop_token_ids", "frequency_penalty", "presence_penalty",
              "repetition_penalty", "min_new_tokens", "skip_special_tokens"):
        if body.get(k) is not None:
            sp[k] = body[k]
    if body.get("stop"):
        sp["stop"] = body["stop"]
    if body.get("n", 1) != 1:
        sp["n"] = body["n"]
    return sp


def _completion_logprobs(meta: dict, base_offset: int = 0) -> Optional[dict]:
    """OpenAI completions `logprobs` object from meta_info (adapter.py to_openai_style_logprobs)."""
    toks = meta.get("output_token_logprobs")
    if toks is None:
        return None
    out = {"tokens": [], "token_logprobs": [], "top_logprobs": [], "text_offset": []}
    off = base_offset
    for (lp, tid, txt), top in zip(toks, meta.get("output_top_logprobs") or [None] * len(toks)):
        piece = txt if txt is not None else str(tid)
        out["tokens"].append(piece)
        out["token_logprobs"].append(lp)
        out["text_offset"].append(off)
        off += len(piece)
        out["top_logprobs"].append({(t if t is not None else str(i)): l for l, i, t in top} if top else None)
    return out


def _chat_logprobs(meta: dict) -> Optional[dict]:
    """OpenAI chat `logprobs.content` list (adapter.py v1_chat_generate_response)."""
    toks = meta.get("output_token_logprobs")
    if toks is None:
        return None
    content = []
    for (lp, tid, txt), top in zip(toks, meta.get("output_top_logprobs") or [None] * len(toks)):
        piece = txt if txt is not None else str(tid)
        content.append({"token": piece, "logprob": lp, "bytes": list(piece.encode("utf-8")),
                        "top_logprobs": [{"token": (t if t is not None else str(i)), "logprob": l,
                                          "bytes": list((t if t is not None else str(i)).encode("utf-8"))}
                                         for l, i, t in (top or [])]})
    return {"content": content}


def build_app(tokenizer_manager, server_args) -> FastAPI:
    app = FastAPI()
    tm = tokenizer_manager
    model_name = server_args.served_model_name

    # ------------------------------------------------------------------ health / info
    @app.get("/health")
    async def health() -> Response:
        return Response(status_code=200)

    @app.get("/health_generate")
    async def health_generate() -> Response:
        """One-token generation round trip through both schedulers (http_server.py:151-196)."""
        obj = {"input_ids": [0], "sampling_params": {"max_new_tokens": 1, "temperature": 0.0}}
        try:
            await asyncio.wait_for(tm.generate_once(obj), timeout=60)
        except Exception as e:  # noqa: BLE001
            return Response(content=str(e), status_code=503)
        return Response(status_code=200)

    @app.get("/get_model_info")
    async def get_model_info():
        return {"model_path": server_args.model_path, "tokenizer_path": server_args.tokenizer_path,
                "is_generation": True}

    @app.get("/get_server_info")
    async def get_server_info():
        import dataclasses
        info = {k: v for k, v in dataclasses.asdict(server_args).items() if k != "model_config"}
        info["model_config"] = type(server_args.model_config).__name__
        info["ready_infos"] = getattr(tm.engine, "ready_infos", [])
        return info

    @app.post("/flush_cache")
    async def flush_cache():
        # Semi-PD forces --disable-radix-cache (server_args.py:325-331): nothing is cached across requests
        return Response(content="Cache flushed.\n", status_code=200)

    @app.get("/v1/models")
    async def available_models():
        return {"object": "list", "data": [{"id": model_name, "object": "model", "created": int(time.time()),
                                            "owned_by": "semi-pd", "root": model_name}]}

    # ------------------------------------------------------------------ native API
    @app.api_route("/generate", methods=["POST", "PUT"])
    async def generate_request(request: Request):
        try:
            obj = await request.json()
        except Exception:  # noqa: BLE001
            return _error("request body is not valid JSON")
        if obj.get("stream"):
            async def stream_results() -> AsyncIterator[bytes]:
                try:
                    async for out in tm.generate_request(obj):
                        yield _sse(out)
                except ValueError as e:
                    yield _sse({"error": {"message": str(e)}})
                yield b"data: [DONE]\n\n"
            return StreamingResponse(stream_results(), media_type="text/event-stream")
        try:
            return await tm.generate_once(obj)
        except ValueError as e:
            return _error(str(e))

    # ------------------------------------------------------------------ OpenAI: completions
    @app.post("/v1/completions")
    async def openai_v1_completions(request: Request):
        try:
            body = await request.json()
        except Exception:  # noqa: BLE001
            return _error("request body is not valid JSON")
        prompt = body.get("prompt")
        if prompt is None:
            return _error("prompt is required")
        obj: Dict[str, Any] = {"sampling_params": _openai_sampling(body, 16), "stream": bool(body.get("stream"))}
        if body.get("logprobs") is not None and body.get("logprobs") is not False:
            obj.update(return_logprob=True, top_logprobs_num=int(body["logprobs"]), return_text_in_logprobs=True)
        if isinstance(prompt, str) or (isinstance(prompt, list) and prompt and isinstance(prompt[0], str)):
            obj["text"] = prompt
        else:
            obj["input_ids"] = prompt
        batch = isinstance(prompt, list) and bool(prompt) and isinstance(prompt[0], (str, list))
        rid = "cmpl-" + uuid.uuid4().hex
        created = int(time.time())
        echo = bool(body.get("echo"))

        n = int(body.get("n", 1) or 1)

        def prompt_text(i: int) -> str:
            if not echo:
                return ""
            p = prompt[i // n] if batch else prompt  # n > 1: choices are prompt-major
            return p if isinstance(p, str) else (tm.tokenizer.decode(p) if tm.tokenizer else "")

        if obj["stream"]:
            async def stream_results() -> AsyncIterator[bytes]:
                sent: Dict[int, int] = {}
                try:
                    async for out in tm.generate_request(obj):
                        i = out.get("index", 0)
                        text = out.get("text", "")
                        start = sent.get(i)
                        delta = text[start:] if start is not None else prompt_text(i) + text
                        sent[i] = len(text)
                        meta = out["meta_info"]
                        final = meta.get("finish_reason") is not None
                        chunk = {"id": rid, "object": "text_completion", "created": created, "model": model_name,
                                 "choices": [{"index": i, "text": delta,
                                              "logprobs": _completion_logprobs(meta) if final else None,
                                              "finish_reason": _openai_finish(meta)}]}
                        if meta.get("finish_reason") is not None:
                            chunk["usage"] = {"prompt_tokens": meta["prompt_tokens"],
                                              "completion_tokens": meta["completion_tokens"],
                                              "total_tokens": meta["prompt_tokens"] + meta["completion_tokens"]}
                        yield _sse(chunk)
                except ValueError as e:
                    yield _sse({"error": {"message": str(e), "type": "BadRequestError", "code": 400}})
                yield b"data: [DONE]\n\n"
            return StreamingResponse(stream_results(), media_type="text/event-stream")
        try:
            ret = await tm.generate_once(obj)
        except ValueError as e:
            return _error(str(e))
        rets = ret if isinstance(ret, list) else [ret]
        choices = [{"index": i, "text": prompt_text(i) + r.get("text", ""),
                    "logprobs": _completion_logprobs(r["meta_info"], len(prompt_text(i))),
                    "finish_reason": _openai_finish(r["meta_info"])} for i, r in enumerate(rets)]
        pt = sum(r["meta_info"]["prompt_tokens"] for r in rets[::n])  # a prompt counts once (adapter.py:560-570)
        ct = sum(r["meta_info"]["completion_tokens"] for r in rets)
        return {"id": rid, "object": "text_completion", "created": created, "model": model_name, "choices": choices,
                "usage": {"prompt_tokens": pt, "completion_tokens": ct, "total_tokens": pt + ct}}

    # ------------------------------------------------------------------ OpenAI: chat
    def chat_prompt_ids(messages: List[dict]) -> List[int]:
        tok = tm.tokenizer
        if tok is None:
            raise ValueError("chat completions need a tokenizer (server runs with --skip-tokenizer-init)")
        if getattr(tok, "chat_template", None):
            ids = tok.apply_chat_template(messages, tokenize=True, add_generation_prompt=True)
            # transformers 4 returns the id list, transformers 5 a BatchEncoding (a Mapping, not a dict)
            return [int(t) for t in (ids["input_ids"] if hasattr(ids, "keys") else ids)]
        text = "".join(f"{m['role']}: {m['content']}\n" for m in messages) + "assistant:"
        return tok.encode(text)

    @app.post("/v1/chat/completions")
    async def openai_v1_chat_completions(request: Request):
        try:
            body = await request.json()
        except Exception:  # noqa: BLE001
            return _error("request body is not valid JSON")
        messages = body.get("messages")
        if not isinstance(messages, list) or not messages:
            return _error("messages is required")
        try:
            ids = chat_prompt_ids(messages)
        except ValueError as e:
            return _error(str(e))
        obj = {"input_ids": ids, "sampling_params": _openai_sampling(body, None), "stream": bool(body.get("stream"))}
        if body.get("logprobs"):
            obj.update(return_logprob=True, top_logprobs_num=int(body.get("top_logprobs") or 0),
                       return_text_in_logprobs=True)
        rid = "chatcmpl-" + uuid.uuid4().hex
        created = int(time.time())
        if obj["stream"]:
            async def stream_results() -> AsyncIterator[bytes]:
                sent_by_index: Dict[int, int] = {}
                try:
                    async for out in tm.generate_request(obj):
                        text, meta, idx = out.get("text", ""), out["meta_info"], out.get("index", 0)
                        if idx not in sent_by_index:
                            yield _sse({"id": rid, "object": "chat.completion.chunk", "created": created,
                                        "model": model_name,
                                        "choices": [{"index": idx, "delta": {"role": "assistant", "content": ""},
                                                     "finish_reason": None}]})
                            sent_by_index[idx] = 0
                        delta, sent_by_index[idx] = text[sent_by_index[idx]:], len(text)
                        chunk = {"id": rid, "object": "chat.completion.chunk", "created": created, "model": model_name,
                                 "choices": [{"index": idx, "delta": {"content": delta},
                                              "logprobs": (_chat_logprobs(meta)
                                                           if meta.get("finish_reason") is not None else None),
                                              "finish_reason": _openai_finish(meta)}]}
                        if meta.get("finish_reason") is not None:
                            chunk["usage"] = {"prompt_tokens": meta["prompt_tokens"],
                                              "completion_tokens": meta["completion_tokens"],
                                              "total_tokens": meta["prompt_tokens"] + meta["completion_tokens"]}
                        yield _sse(chunk)
                except ValueError as e:
                    yield _sse({"error": {"message": str(e), "type": "BadRequestError", "code": 400}})
                yield b"data: [DONE]\n\n"
            return StreamingResponse(stream_results(), media_type="text/event-stream")
        try:
            r = await tm.generate_once(obj)
        except ValueError as e:
            return _error(str(e))
        rets = r if isinstance(r, list) else [r]  # n > 1: one result per sample
        pt = rets[0]["meta_info"]["prompt_tokens"]
        ct = sum(x["meta_info"]["completion_tokens"] for x in rets)
        return {"id": rid, "object": "chat.completion", "created": created, "model": model_name,
                "choices": [{"index": i, "message": {"role": "assistant", "content": x.get("text", "")},
                             "logprobs": _chat_logprobs(x["meta_info"]), "finish_reason": _openai_finish(x["meta_info"])}
                            for i, x in enumerate(rets)],
                "usage": {"prompt_tokens": pt, "completion_tokens": ct, "total_tokens": pt + ct}}

    @app.on_event("shutdown")
    def _shutdown():
        tm.shutdown()
        tm.engine.shutdown()

    return app


def launch_server(server_args, tokenizer=None):
    """http_server.py:602-675 launch_server: start the engine (scheduler processes), the tokenizer
    manager and serve."""
    import uvicorn

    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.tokenizer_manager import TokenizerManager, get_tokenizer

    if tokenizer is None and not server_args.skip_tokenizer_init:
        tokenizer = get_tokenizer(server_args.tokenizer_path)
    if tokenizer is not None and server_args.eos_token_ids is None and tokenizer.eos_token_id is not None:
        server_args.eos_token_ids = [int(tokenizer.eos_token_id)]
    engine = Engine(server_args)
    tm = TokenizerManager(engine, server_args, tokenizer=tokenizer)
    app = build_app(tm, server_args)
    logger.info("The server is fired up and ready to roll!")
    uvicorn.run(app, host=server_args.host, port=server_args.port, log_level="info", timeout_keep_alive=5)
