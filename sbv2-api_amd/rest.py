"""The REST surface of crates/sbv2_api/src/main.rs over a TTSModelHolder (holder.py): same routes, request schema, defaults, content types
and error mapping, so that a client of the reference's server cannot tell the difference.

  GET  /            "Hello, World!"                                              main.rs:193
  GET  /models      JSON list of idents                                          main.rs:24-33
  POST /synthesize  {text, ident, sdp_ratio = 0.0, length_scale = 1.0, style_id = 0, speaker_id = 0} -> audio/wav     main.rs:51-100
  any error         500 text/plain "Something went wrong: <message>"            sbv2_api/src/error.rs:10-18
  one request at a time (Arc<Mutex<TTSModelHolder>>, main.rs:86,104)             -> a lock around the holder

FastAPI / starlette are plumbing here; `python -m sbv2_api_amd.rest` is not provided on purpose: a deployment needs the text front end
(G2P + tokenizer, out of scope: SURVEY.md §2 #7-12) plugged into the holder's `parse_text`."""
import threading


def make_app(holder):
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse, PlainTextResponse, Response
    from pydantic import BaseModel

    from . import orchestrator

    class SynthesizeRequest(BaseModel):      # main.rs:51-63
        text: str
        ident: str
        sdp_ratio: float = 0.0
        length_scale: float = 1.0
        style_id: int = 0
        speaker_id: int = 0

    app = FastAPI(docs_url="/docs")          # main.rs:196 serves the OpenAPI document at /docs as well
    lock = threading.Lock()

    @app.exception_handler(Exception)
    async def _err(_: Request, exc: Exception):
        return PlainTextResponse(f"Something went wrong: {exc}", status_code=500)

    @app.get("/", response_class=PlainTextResponse)
    def root():
        return "Hello, World!"

    @app.get("/models")
    def models():
        with lock:
            return JSONResponse(holder.models())

    @app.post("/synthesize")
    def synthesize(req: SynthesizeRequest):
        try:
            with lock:
                wav = holder.easy_synthesize(req.ident, req.text, req.style_id, req.speaker_id,
                                             orchestrator.SynthesizeOptions(sdp_ratio=req.sdp_ratio, length_scale=req.length_scale))
        except Exception as e:                # any error -> 500 + text, like AppError::into_response
            return PlainTextResponse(f"Something went wrong: {e}", status_code=500)
        return Response(content=wav, media_type="audio/wav")

    return app
