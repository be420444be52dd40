"""Host mirror of `sbv2_core::tts::TTSModelHolder` (crates/sbv2_core/src/tts.rs:40-349) over the HIP path: the model cache whose sessions are
GPU-resident weight sets instead of ort::Sessions (SURVEY.md §8f row 3, §5 "checkpoint / resume": loaded = weights in HBM, evicted = freed).

Reference behaviour kept, quirks included:
  new(bert, tokenizer, max_loaded_models)   tts.rs:56-71    DeBERTa session loaded once; tokenizer / G2P are the front end (out of scope,
                                                            SURVEY.md §2 #7-12): here the caller supplies `parse_text`, the same seam the wasm build uses
  models()                                  tts.rs:74-76
  load(ident, style_vectors, vits2)         tts.rs:149-179  ignored when the ident exists; the session is created only while fewer than
                                                            max_loaded_models are resident; the raw bytes are kept iff max_loaded_models is set
  load_sbv2file(ident, bytes)               tts.rs:132-140  parse_sbv2file -> load
  unload(ident)                             tts.rs:182-196  REMOVES the entry (bytes included)
  find_and_load_model(ident)                tts.rs:223-258  a non-resident model is re-created from its kept bytes; when the cache is full the FIRST
                                                            entry of the list is unloaded, i.e. dropped from the holder altogether (the reference does
                                                            exactly that; it is not an LRU)
  get_style_vector                          tts.rs:264-271 -> style.rs:19-28
  easy_synthesize(ident, text, ...)         tts.rs:280-349  split on '\\n', skip empty lines, 22050-sample gaps, noise 0.677 / 0.8, WAV
What differs on purpose: the sentences of a request go through ONE batched pipeline call (orchestrator.easy_synthesize).
"""
import ctypes as C

from . import _lib, model, orchestrator


class ModelNotFoundError(model.Sbv2Error):
    """sbv2_core::error::Error::ModelNotFoundError (error.rs:24-25)."""


def parse_sbv2file(sbv2_bytes: bytes):
    """sbv2file.rs:15-37 through the C ABI -> (style_vectors_bytes, vits2_bytes)."""
    l = _lib.lib()
    a, an, b, bn = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
    buf = (C.c_char * len(sbv2_bytes)).from_buffer_copy(sbv2_bytes)
    _lib.check(l.sbv2_parse_sbv2file(C.cast(buf, C.c_void_p), len(sbv2_bytes), C.byref(a), C.byref(an), C.byref(b), C.byref(bn)))
    try:
        return C.string_at(a, an.value), C.string_at(b, bn.value)
    finally:
        l.sbv2_bytes_free(a)
        l.sbv2_bytes_free(b)


def aivmx_style_vectors(aivmx_bytes: bytes):
    """tts.rs:92-108 through the C ABI: the style table stored in an .aivmx (= ONNX) file's metadata -> float32 [n, dim]."""
    import numpy as np
    l = _lib.lib()
    d, n, dim = C.c_void_p(), C.c_int64(), C.c_int64()
    buf = (C.c_char * len(aivmx_bytes)).from_buffer_copy(aivmx_bytes)
    _lib.check(l.sbv2_aivmx_style_vectors(C.cast(buf, C.c_void_p), len(aivmx_bytes), C.byref(d), C.byref(n), C.byref(dim)))
    try:
        return np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_float)), shape=(n.value, dim.value)).copy()
    finally:
        l.sbv2_bytes_free(d)


class _TTSModel:
    """tts.rs:32-38"""

    def __init__(self, ident, vits2, style_vectors, raw):
        self.ident, self.vits2, self.style_vectors, self.bytes = ident, vits2, style_vectors, raw
        self.pipe = None


class TTSModelHolder:
    def __init__(self, bert_model_bytes: bytes, parse_text=None, max_loaded_models=None, device: int = 0, load_session=None, make_pipeline=None):
        """parse_text(sentence: str) -> {input_ids, word2ph, phones, tones, langs}: the text front end (tts_util.rs:93-160 minus the BERT call,
        which happens on the device inside the pipeline).  load_session / make_pipeline are injection points for CPU tests."""
        self._load_session = load_session or (lambda b, is_bert: model.load_model(b, is_bert, device=device))
        self._make_pipeline = make_pipeline or model.Pipeline
        self.bert = self._load_session(bert_model_bytes, True)
        self.parse_text = parse_text
        self.max_loaded_models = max_loaded_models
        self.models_ = []

    # ---- tts.rs:74-76
    def models(self):
        return [m.ident for m in self.models_]

    def _resident(self):
        return sum(1 for m in self.models_ if m.vits2 is not None)

    def _find(self, ident):
        for m in self.models_:
            if m.ident == ident:
                return m
        raise ModelNotFoundError(f"model not found: {ident}")

    # ---- tts.rs:149-179
    def load(self, ident: str, style_vectors_bytes: bytes, vits2_bytes: bytes):
        try:
            self._find(ident)
            return
        except ModelNotFoundError:
            pass
        do_load = self.max_loaded_models is None or self._resident() < self.max_loaded_models
        sess = self._load_session(vits2_bytes, False) if do_load else None
        sv = orchestrator.load_style(style_vectors_bytes)
        self.models_.append(_TTSModel(ident, sess, sv, bytes(vits2_bytes) if self.max_loaded_models is not None else None))

    # ---- tts.rs:77-130 (cargo feature "aivmx"): same cache rules as load(); the style table comes out of the model file itself
    def load_aivmx(self, ident: str, aivmx_bytes: bytes):
        try:
            self._find(ident)
            return
        except ModelNotFoundError:
            pass
        do_load = self.max_loaded_models is None or self._resident() < self.max_loaded_models
        sv = aivmx_style_vectors(aivmx_bytes)
        sess = self._load_session(aivmx_bytes, False) if do_load else None
        self.models_.append(_TTSModel(ident, sess, sv, bytes(aivmx_bytes) if self.max_loaded_models is not None else None))

    # ---- tts.rs:132-140
    def load_sbv2file(self, ident: str, sbv2_bytes: bytes):
        style, vits2 = parse_sbv2file(sbv2_bytes)
        self.load(ident, style, vits2)

    # ---- tts.rs:182-196
    def unload(self, ident: str) -> bool:
        for i, m in enumerate(self.models_):
            if m.ident == ident:
                self._drop(m)
                del self.models_[i]
                return True
        return False

    @staticmethod
    def _drop(m):
        if m.pipe is not None:
            m.pipe.close()
            m.pipe = None
        if m.vits2 is not None:
            m.vits2.close()      # frees the model's weights and workspace in HBM
            m.vits2 = None

    # ---- tts.rs:223-258
    def find_and_load_model(self, ident: str) -> bool:
        m = self._find(ident)
        if m.vits2 is not None:
            return True
        raw, sv = m.bytes, m.style_vectors
        self.unload(ident)
        sess = self._load_session(raw, False)
        if self.max_loaded_models is not None and self._resident() >= self.max_loaded_models:
            self.unload(self.models_[0].ident)       # the reference removes the first entry of the Vec, resident or not
        self.models_.append(_TTSModel(ident, sess, sv, raw))
        return True

    # ---- tts.rs:264-271
    def get_style_vector(self, ident: str, style_id: int, weight: float):
        return orchestrator.get_style_vector(self._find(ident).style_vectors, style_id, weight)

    # ---- tts.rs:280-349
    def easy_synthesize(self, ident: str, text, style_id: int = 0, speaker_id: int = 0, options=None, noise_seed=None) -> bytes:
        """`text`: a str (needs parse_text) or the already parsed list of sentence dicts / None for empty lines."""
        options = options or orchestrator.SynthesizeOptions()
        self.find_and_load_model(ident)
        m = self._find(ident)
        if isinstance(text, str):
            if self.parse_text is None:
                raise model.Sbv2Error("no text front end configured (parse_text): pass parsed sentences instead")
            lines = text.split("\n") if options.split_sentences else [text]
            sentences = [self.parse_text(t) if t else None for t in lines]
        else:
            sentences = list(text)
        if m.pipe is None:
            m.pipe = self._make_pipeline(self.bert, m.vits2)
        return orchestrator.easy_synthesize(m.pipe, sentences, m.style_vectors, style_id, speaker_id, options, noise_seed=noise_seed)

    def close(self):
        for m in self.models_:
            self._drop(m)
        self.models_ = []
        if self.bert is not None:
            self.bert.close()
            self.bert = None
