"""Drop-in call surface of ``transformers.pipeline("automatic-speech-recognition", ...)`` for the
CrisperWhisper word-timestamp path (REF/transcribe.py:21-33, REF/app.py:51-61,102).

    pipe = crisperwhisper_amd.pipeline("automatic-speech-recognition", model=model, tokenizer=tok,
                                       feature_extractor=fe, chunk_length_s=30, batch_size=16,
                                       return_timestamps="word", torch_dtype=dtype, device="cuda:0")
    result = pipe(path_or_array)     # {"text": str, "chunks": [{"text", "timestamp": (start, end)}]}

Same argument names, accepted input forms, error types and output schema as
``AutomaticSpeechRecognitionPipeline`` (TF/pipelines/automatic_speech_recognition.py:190-247, 345-481,
600-710); the device work goes through libcrisperwhisper.so.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional

import numpy as np

from . import audio, collate, dist, generation, utils
from ._native import N_SAMPLES
from .engine import Engine, ModelSpec

logger = logging.getLogger("crisperwhisper_amd")
DEFAULT_NUM_BEAMS = 5       # TF/pipelines/automatic_speech_recognition.py:160-163 (5.x pipeline default)
_warned = set()


def _warn_once(key: str, msg: str):
    if key not in _warned:
        _warned.add(key)
        logger.warning(msg)


class ModelBundle:
    """Geometry + generation settings + weights (HF state_dict names -> float32 numpy)."""

    def __init__(self, spec: ModelSpec, weights: Dict[str, np.ndarray]):
        self.spec = spec
        self.weights = weights

    @classmethod
    def from_hf(cls, model) -> "ModelBundle":
        cfg, gc = model.config, model.generation_config
        if not hasattr(gc, "alignment_heads"):
            raise ValueError("Model generation config has no `alignment_heads`, token-level timestamps not available. "
                             "See https://gist.github.com/hollance/42e32852f24243b748ae6bc1f985b13a on how to add this "
                             "property to the generation config.")
        if not hasattr(gc, "no_timestamps_token_id"):
            raise ValueError("The generation config is outdated: `no_timestamps_token_id` is missing.")
        spec = ModelSpec(
            d_model=cfg.d_model, n_heads=cfg.encoder_attention_heads, ffn_dim=cfg.encoder_ffn_dim,
            enc_layers=cfg.encoder_layers, dec_layers=cfg.decoder_layers, n_mels=cfg.num_mel_bins,
            vocab_size=cfg.vocab_size, max_target_positions=cfg.max_target_positions,
            median_filter_width=cfg.median_filter_width,
            alignment_heads=[list(h) for h in gc.alignment_heads],
            eos_token_id=gc.eos_token_id if isinstance(gc.eos_token_id, int) else gc.eos_token_id[0],
            pad_token_id=gc.pad_token_id, decoder_start_token_id=gc.decoder_start_token_id,
            no_timestamps_token_id=gc.no_timestamps_token_id,
            max_initial_timestamp_index=getattr(gc, "max_initial_timestamp_index", None),
            suppress_tokens=list(gc.suppress_tokens or []), begin_suppress_tokens=list(gc.begin_suppress_tokens or []),
            lang_to_id=dict(getattr(gc, "lang_to_id", {}) or {}), task_to_id=dict(getattr(gc, "task_to_id", {}) or {}),
            max_length=gc.max_length or cfg.max_target_positions,
            forced_decoder_ids=(getattr(gc, "forced_decoder_ids", None) if getattr(gc, "forced_decoder_ids", None) is not None
                                else getattr(cfg, "forced_decoder_ids", None)),
            language=getattr(gc, "language", None), task=getattr(gc, "task", None))
        weights = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items() if k != "proj_out.weight"}
        return cls(spec, weights)

    @classmethod
    def from_pretrained(cls, path: str) -> "ModelBundle":
        """Load a Whisper checkpoint directory without ``transformers``: ``config.json`` + ``generation_config.json``
        -> ModelSpec, ``model.safetensors`` (or the sharded ``model.safetensors.index.json``) -> weights, streamed tensor
        by tensor as float32.  Replaces ``AutoModelForSpeechSeq2Seq.from_pretrained(model_id)`` (REF/transcribe.py:14-17)
        for a local snapshot; error texts follow the reference's (``generation_whisper.py:1399-1405, 1689-1693``)."""
        import json
        import os
        cfg = json.load(open(os.path.join(path, "config.json")))
        gpath = os.path.join(path, "generation_config.json")
        gc = json.load(open(gpath)) if os.path.exists(gpath) else {}
        if cfg.get("model_type", "whisper") != "whisper":
            raise ValueError(f"not a Whisper checkpoint: model_type={cfg.get('model_type')!r}")
        if "alignment_heads" not in gc:
            raise ValueError("Model generation config has no `alignment_heads`, token-level timestamps not available. "
                             "See https://gist.github.com/hollance/42e32852f24243b748ae6bc1f985b13a on how to add this "
                             "property to the generation config.")
        if "no_timestamps_token_id" not in gc:
            raise ValueError("The generation config is outdated: `no_timestamps_token_id` is missing.")
        eos = gc.get("eos_token_id", cfg.get("eos_token_id"))
        spec = ModelSpec(
            d_model=cfg["d_model"], n_heads=cfg["encoder_attention_heads"], ffn_dim=cfg["encoder_ffn_dim"],
            enc_layers=cfg["encoder_layers"], dec_layers=cfg["decoder_layers"], n_mels=cfg["num_mel_bins"],
            vocab_size=cfg["vocab_size"], max_target_positions=cfg.get("max_target_positions", 448),
            median_filter_width=cfg.get("median_filter_width", 7),
            alignment_heads=[list(h) for h in gc["alignment_heads"]],
            eos_token_id=eos if isinstance(eos, int) else eos[0],
            pad_token_id=gc.get("pad_token_id", cfg.get("pad_token_id")),
            decoder_start_token_id=gc.get("decoder_start_token_id", cfg.get("decoder_start_token_id")),
            no_timestamps_token_id=gc["no_timestamps_token_id"],
            max_initial_timestamp_index=gc.get("max_initial_timestamp_index"),
            suppress_tokens=list(gc.get("suppress_tokens") or []), begin_suppress_tokens=list(gc.get("begin_suppress_tokens") or []),
            lang_to_id=dict(gc.get("lang_to_id") or {}), task_to_id=dict(gc.get("task_to_id") or {}),
            max_length=gc.get("max_length") or cfg.get("max_target_positions", 448),
            forced_decoder_ids=gc.get("forced_decoder_ids") if gc.get("forced_decoder_ids") is not None else cfg.get("forced_decoder_ids"),
            language=gc.get("language"), task=gc.get("task"))
        return cls(spec, _SafetensorsWeights(path))


    @classmethod
    def from_ctranslate2(cls, path: str) -> "ModelBundle":
        """Load a faster-whisper / CTranslate2 model directory (``model.bin`` + ``config.json`` + ``tokenizer.json`` or
        ``vocabulary.json``; REF/README.md:186-203 distributes CrisperWhisper in this form too).  Geometry comes from the
        tensor shapes and the recorded head count, the generation settings from CTranslate2's ``config.json``
        (``alignment_heads``, ``suppress_ids``, ``suppress_ids_begin``, ``lang_ids``) and the token table; weights are
        de-fused / de-quantised to the transformers names the engine loads.  Layout restated from CTranslate2's published
        spec -- parity unpinned, see ``crisperwhisper_amd/ct2.py``."""
        import json
        import os
        from . import ct2
        from .languages import LANGUAGES
        _, var, aliases = ct2.read_model_bin(os.path.join(path, "model.bin"))
        geo = ct2.geometry(var)
        cpath = os.path.join(path, "config.json")
        cfg = json.load(open(cpath)) if os.path.exists(cpath) else {}
        if not cfg.get("alignment_heads"):
            raise ValueError("Model generation config has no `alignment_heads`, token-level timestamps not available. "
                             "See https://gist.github.com/hollance/42e32852f24243b748ae6bc1f985b13a on how to add this "
                             "property to the generation config.")
        ids = ct2.vocabulary_ids(path)
        if "<|notimestamps|>" not in ids:
            raise ValueError("The generation config is outdated: `no_timestamps_token_id` is missing.")
        lang_to_id = {f"<|{c}|>": ids[f"<|{c}|>"] for c in LANGUAGES if f"<|{c}|>" in ids}
        spec = ModelSpec(
            d_model=geo["d_model"], n_heads=geo["n_heads"], ffn_dim=geo["ffn_dim"], enc_layers=geo["enc_layers"],
            dec_layers=geo["dec_layers"], n_mels=geo["n_mels"], vocab_size=geo["vocab_size"],
            max_target_positions=geo["max_target_positions"], median_filter_width=7,
            alignment_heads=[list(h) for h in cfg["alignment_heads"]],
            eos_token_id=ids["<|endoftext|>"], pad_token_id=ids["<|endoftext|>"],
            decoder_start_token_id=ids["<|startoftranscript|>"], no_timestamps_token_id=ids["<|notimestamps|>"],
            max_initial_timestamp_index=50,
            suppress_tokens=[int(t) for t in cfg.get("suppress_ids", []) if int(t) >= 0],
            begin_suppress_tokens=[int(t) for t in cfg.get("suppress_ids_begin", []) if int(t) >= 0],
            lang_to_id=lang_to_id,
            task_to_id={k: ids[f"<|{k}|>"] for k in ("transcribe", "translate") if f"<|{k}|>" in ids},
            max_length=geo["max_target_positions"], forced_decoder_ids=None, language=None, task=None)
        return cls(spec, ct2.to_hf_state(var, aliases))


class _SafetensorsWeights(dict):
    """``items()`` streams (HF name, float32 array) out of model.safetensors / its shards; nothing is held in memory."""

    def __init__(self, path: str):
        super().__init__()
        import json
        import os
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(idx):
            files = sorted(set(json.load(open(idx))["weight_map"].values()))
        else:
            files = ["model.safetensors"]
        self.files = [os.path.join(path, f) for f in files]
        for f in self.files:
            if not os.path.exists(f):
                raise FileNotFoundError(f"{f} not found (only safetensors checkpoints are read natively)")

    def items(self):
        from safetensors import safe_open
        for fn in self.files:
            try:
                with safe_open(fn, framework="np") as f:
                    for k in f.keys():
                        if k != "proj_out.weight":
                            yield k, np.ascontiguousarray(f.get_tensor(k), dtype=np.float32)
            except TypeError:          # bfloat16 has no numpy dtype: go through torch for the cast
                with safe_open(fn, framework="pt") as f:
                    for k in f.keys():
                        if k != "proj_out.weight":
                            yield k, f.get_tensor(k).float().numpy()

    def __bool__(self):
        return True


def _dtype_name(dtype) -> str:
    if dtype is None:
        _warn_once("dtype", "no torch_dtype given: running the bf16 engine (HF would keep the checkpoint dtype)")
        return "bf16"
    s = str(dtype)
    if "float32" in s or s in ("f32", "fp32"):
        return "f32"
    if ("float16" in s and "bfloat16" not in s) or s in ("f16", "fp16", "half"):
        return "f16"       # the reference's GPU dtype (REF/transcribe.py:10): the binary16 build of the MFMA engine
    return "bf16"


def _device_index(device) -> int:
    if device is None:
        return 0
    if isinstance(device, int):
        if device < 0:
            raise ValueError("crisperwhisper_amd has no CPU path: device must be a GPU (e.g. 'cuda:0')")
        return device
    s = str(device)
    if s == "cpu":
        raise ValueError("crisperwhisper_amd has no CPU path: device must be a GPU (e.g. 'cuda:0')")
    return int(s.split(":")[1]) if ":" in s else 0


# generate_kwargs the native path implements (TF generation_whisper.py:generate); everything else raises instead of being
# dropped: a drop-in must either honour an argument or refuse it
_GENERATE_KWARGS = ("language", "task", "max_new_tokens", "min_new_tokens", "num_beams", "length_penalty", "early_stopping",
                    "do_sample", "temperature", "num_return_sequences", "prompt_ids", "assistant_model",
                    "logprob_threshold", "no_speech_threshold", "compression_ratio_threshold", "return_timestamps")


def _check_generate_kwargs(gk: Dict[str, Any], default_num_beams: Optional[int] = None) -> None:
    """``default_num_beams``: the width a call without ``num_beams`` decodes with (the pipeline default): the refusals that
    depend on the beam width are then raised here, before any audio is loaded."""
    unknown = sorted(k for k in gk if k not in _GENERATE_KWARGS)
    if unknown:
        raise ValueError(f"generate_kwargs {unknown} are not implemented on the native path (implemented: "
                         f"{', '.join(_GENERATE_KWARGS)}); they would be silently ignored otherwise")
    thresholds = [k for k in ("compression_ratio_threshold", "logprob_threshold", "no_speech_threshold") if gk.get(k) is not None]
    if gk.get("do_sample"):
        raise ValueError("generate_kwargs['do_sample'] is not supported on the native path (deterministic greedy / beam search "
                         "only: stochastic decoding cannot be made bit-comparable with torch's generator)")
    temp = gk.get("temperature")
    if isinstance(temp, (tuple, list)):
        # temperature fallback (generation_whisper.py:970-1116) only fires when a threshold is set
        if thresholds or len(temp) == 0:
            raise ValueError("temperature fallback with sampling is not implemented on the native path "
                             "(stochastic decoding cannot be made bit-comparable with torch's generator)")
        temp = temp[0]
    # do_sample = temperature > 0.0 (generation_whisper.py:1002): any positive temperature, 1.0 included, makes transformers
    # SAMPLE (with num_beams forced to 1) -- only 0 is the deterministic decoding this path implements
    if temp is not None and not (isinstance(temp, (int, float)) and not isinstance(temp, bool) and float(temp) == 0.0):
        raise ValueError(f"generate_kwargs['temperature']={gk['temperature']!r} is not supported on the native path: transformers "
                         "samples at every temperature > 0 (stochastic decoding cannot be made bit-comparable with torch's "
                         "generator); pass temperature=0.0 or leave it out")
    if gk.get("num_return_sequences") not in (None, 1):
        raise ValueError("generate_kwargs['num_return_sequences'] > 1 is not supported on the native path")
    for k in ("prompt_ids", "assistant_model"):
        if gk.get(k) is not None:
            raise ValueError(f"generate_kwargs[{k!r}] is not supported on the native path")
    if gk.get("return_timestamps") not in (None, True, "word"):
        raise ValueError("generate_kwargs['return_timestamps'] must be left to the pipeline argument of the same name")
    if gk.get("no_speech_threshold") is not None and gk.get("logprob_threshold") is None:
        raise ValueError("no_speech_threshold needs logprob_threshold as well (generation_whisper.py:1275-1285 compares both)")
    beams = gk.get("num_beams") if gk.get("num_beams") is not None else default_num_beams
    if thresholds and beams not in (None, 1) and any(k in thresholds for k in ("logprob_threshold", "no_speech_threshold")):
        raise ValueError(f"logprob_threshold / no_speech_threshold are implemented for greedy decoding only and this call decodes "
                         f"with {beams} beams" + ("" if gk.get("num_beams") is not None else " (the pipeline default)") +
                         ": pass generate_kwargs={'num_beams': 1, ...} -- transformers scores beam hypotheses differently again "
                         "and that path is not reproduced")
    if gk.get("logprob_threshold") is not None and gk.get("temperature") is None:
        raise ValueError("logprob_threshold needs an explicit temperature (pass temperature=0.0): transformers itself fails with "
                         "a TypeError in _retrieve_avg_logprobs otherwise (generation_whisper.py:1959)")
    # with one temperature a failed compression-ratio / log-probability check has nowhere to fall back to and HF keeps the
    # result (generation_whisper.py:1100-1104): the only observable effect of the thresholds at temperature 0 is the no-speech
    # skip, which generation.generate implements; compression_ratio_threshold is therefore accepted and has no effect


class CrisperWhisperPipeline:
    def __init__(self, model, tokenizer=None, feature_extractor=None, chunk_length_s=0, stride_length_s=None,
                 batch_size=1, return_timestamps=None, torch_dtype=None, dtype=None, device=None,
                 shard: Optional[dist.Shard] = None, contexts: int = 1, cross_kv_dtype: Optional[str] = None,
                 encoder_gemm_dtype: Optional[str] = None,
                 engines: Optional[List[Engine]] = None, num_beams: Optional[int] = None, **kwargs):
        """``num_beams`` (construction time): the widest beam the contexts are provisioned for -- decoder rows =
        batch_size x num_beams, at most 64.  Default 5 = ``AutomaticSpeechRecognitionPipeline._default_generation_config``
        of the installed transformers (TF/pipelines/automatic_speech_recognition.py:160-163), which is what a call
        without ``generate_kwargs`` runs (REF/transcribe.py:33); pass ``generate_kwargs={"num_beams": 1}`` per call for
        the greedy decoding of the 2024 reference."""
        if isinstance(model, str):               # local checkpoint directory: no transformers object needed
            if tokenizer is None:
                tokenizer = collate.Vocabulary.from_pretrained(model)
            import os as _os
            # a faster-whisper / CTranslate2 directory (model.bin, no safetensors) is read through its own loader
            ct2_dir = _os.path.exists(_os.path.join(model, "model.bin")) and not any(
                _os.path.exists(_os.path.join(model, f)) for f in ("model.safetensors", "model.safetensors.index.json"))
            model = ModelBundle.from_ctranslate2(model) if ct2_dir else ModelBundle.from_pretrained(model)
        self.bundle = model if isinstance(model, ModelBundle) else ModelBundle.from_hf(model)
        if tokenizer is None:
            raise ValueError("a tokenizer (WhisperTokenizer or crisperwhisper_amd.collate.Vocabulary) is required")
        if isinstance(tokenizer, str):
            tokenizer = collate.Vocabulary.from_pretrained(tokenizer)
        self.vocab = tokenizer if isinstance(tokenizer, collate.Vocabulary) else collate.Vocabulary.from_hf_tokenizer(tokenizer)
        self.sampling_rate = getattr(feature_extractor, "sampling_rate", audio.SAMPLING_RATE)
        if feature_extractor is not None and getattr(feature_extractor, "feature_size", self.bundle.spec.n_mels) != self.bundle.spec.n_mels:
            raise ValueError("feature_extractor.feature_size does not match model.config.num_mel_bins")
        self.chunk_length_s = chunk_length_s
        self.stride_length_s = stride_length_s
        self.batch_size = int(batch_size or 1)
        self.default_num_beams = DEFAULT_NUM_BEAMS
        self.max_rows = min(64, self.batch_size * int(num_beams or DEFAULT_NUM_BEAMS))
        self.max_rows = max(self.max_rows, min(64, self.batch_size))
        self.return_timestamps = return_timestamps
        self.shard = shard or dist.Shard()
        # `contexts` > 1: independent engine contexts on the same GPU, each running its own batches from a host
        # thread -- the decode chain is latency-bound, so a second in-flight batch fills idle CUs.
        if engines:                                # already created and loaded by the caller (bench.py shares them)
            self.engines = list(engines)
            if any(e.max_batch < self.batch_size for e in self.engines):
                raise ValueError("engines were created with a smaller max_batch than batch_size")
            self.max_rows = min(e.max_batch for e in self.engines)
        else:
            self.engines = [Engine(self.bundle.spec, dtype=_dtype_name(dtype if dtype is not None else torch_dtype),
                                   max_batch=self.max_rows, device=_device_index(device), cross_kv_dtype=cross_kv_dtype,
                                   encoder_gemm_dtype=encoder_gemm_dtype)
                            for _ in range(max(1, int(contexts)))]
            for e in self.engines:
                e.load_state_dict(self.bundle.weights)
        self.engine = self.engines[0]
        utils.bind_engine(self.engine)
        self.stats: Dict[str, Any] = {}

    # -- input forms (TF/pipelines/automatic_speech_recognition.py:345-430) ------------------------
    def _load(self, inputs) -> np.ndarray:
        if isinstance(inputs, str):
            if inputs.startswith("http://") or inputs.startswith("https://"):
                raise ValueError("remote URLs are not fetched by the native pipeline; pass a local path or an array")
            inputs = audio.read_audio(inputs, self.sampling_rate, self.engine)
        elif isinstance(inputs, bytes):
            inputs = audio.decode_wav_bytes(inputs, self.sampling_rate, self.engine)
        if hasattr(inputs, "detach") and hasattr(inputs, "cpu"):       # torch.Tensor
            inputs = inputs.detach().cpu().numpy()
        if isinstance(inputs, dict):
            inputs = dict(inputs)
            inputs.pop("stride", None)
            if not ("sampling_rate" in inputs and ("raw" in inputs or "array" in inputs)):
                raise ValueError(
                    "When passing a dictionary to AutomaticSpeechRecognitionPipeline, the dict needs to contain a "
                    '"raw" key containing the numpy array or torch tensor representing the audio and a "sampling_rate" key, '
                    "containing the sampling_rate associated with that array")
            arr = inputs.pop("raw", None)
            if arr is None:
                arr = inputs.pop("array", None)
            if hasattr(arr, "detach"):
                arr = arr.detach().cpu().numpy()
            inputs = audio.resample(np.asarray(arr, dtype=np.float32), int(inputs["sampling_rate"]), self.sampling_rate, self.engine)
        if not isinstance(inputs, np.ndarray):
            raise TypeError(f"We expect a numpy ndarray or torch tensor as input, got `{type(inputs)}`")
        if inputs.ndim != 1:
            logger.warning("We expect a single channel audio input for AutomaticSpeechRecognitionPipeline, got %d. "
                           "Taking the mean of the channels for mono conversion.", inputs.ndim)
            inputs = inputs.mean(axis=0)
        return np.ascontiguousarray(inputs, dtype=np.float32)

    def __call__(self, inputs, **kwargs):
        if isinstance(inputs, (list, tuple)):
            return [self._run_one(x, **kwargs) for x in inputs]
        return self._run_one(inputs, **kwargs)

    def _run_one(self, inputs, return_timestamps=None, generate_kwargs=None, chunk_length_s=None,
                 stride_length_s=None, return_language=None, **unused):
        rt = return_timestamps if return_timestamps is not None else self.return_timestamps
        if not (rt == "word" or rt is True):
            raise ValueError("crisperwhisper_amd implements the timestamped paths: pass return_timestamps='word' "
                             "(CrisperWhisper's purpose, REF/transcribe.py:28) or True (segment-level chunks, the "
                             "setting REF/app.py:51-61 constructs its pipeline with)")
        if return_language:
            raise ValueError("return_language is not supported on the native path")
        gk = dict(generate_kwargs or {})
        _check_generate_kwargs(gk, self.default_num_beams)
        if "num_beams" not in gk:
            _warn_once("beams", f"no num_beams given: decoding with {self.default_num_beams} beams like the installed transformers "
                                "ASR pipeline default; pass generate_kwargs={'num_beams': 1} for the greedy decoding of the 2024 reference")
        num_beams = int(gk.get("num_beams", self.default_num_beams))
        if gk.get("length_penalty") not in (None, 1.0) or gk.get("early_stopping") not in (None, False):
            raise ValueError("only the default length_penalty=1.0 / early_stopping=False beam search is implemented")
        import time as _time
        t_ph = [_time.perf_counter()]                       # phase clock of this call: load | local batches | gather | collation
        pcm = self._load(inputs)
        t_ph.append(_time.perf_counter())
        cl = self.chunk_length_s if chunk_length_s is None else chunk_length_s
        sl = self.stride_length_s if stride_length_s is None else stride_length_s
        sr = self.sampling_rate
        if cl:
            if sl is None:
                sl = cl / 6
            if isinstance(sl, (int, float)):
                sl = [sl, sl]
            chunk_len = int(round(cl * sr))
            windows = audio.chunk_windows(len(pcm), chunk_len, int(round(sl[0] * sr)), int(round(sl[1] * sr)))
            if chunk_len > N_SAMPLES:
                raise ValueError("chunk_length_s must be <= 30 for Whisper")
            with_stride = True
        else:
            if len(pcm) > N_SAMPLES:
                raise ValueError("audio longer than 30 s needs chunk_length_s=30 (the reference call, REF/transcribe.py:26); "
                                 "sequential long-form decoding is not implemented on the native path")
            windows = [(0, len(pcm), (len(pcm), 0, 0), True)]
            with_stride = False

        lo, hi = dist.shard_bounds(len(windows), self.shard.world)[self.shard.rank]
        mine = list(range(lo, hi))
        def run_batch(args):
            slot, idxs = args
            eng = self.engines[slot % len(self.engines)]
            clips = [pcm[windows[i][0]: windows[i][0] + windows[i][1]] for i in idxs]
            _, nf = eng.mel(clips)
            st = {}
            out = generation.generate(
                eng, len(idxs), nf, language=gk.get("language"), task=gk.get("task"),
                max_new_tokens=gk.get("max_new_tokens"), min_new_tokens=gk.get("min_new_tokens"),
                num_beams=num_beams, stats=st, logprob_threshold=gk.get("logprob_threshold"),
                no_speech_threshold=gk.get("no_speech_threshold"))
            rs = []
            for k, i in enumerate(idxs):
                n_tok = len(out["token_timestamps"][k])
                stride = tuple(x / sr for x in windows[i][2])
                rs.append(dist.pack_record(i, out["sequences"][k][:n_tok], out["token_timestamps"][k], stride))
            return rs, st.get("generate_calls", 0)

        per = self.batch_size
        if num_beams > 1:
            if num_beams > self.max_rows:
                raise ValueError(f"num_beams={num_beams} exceeds the {self.max_rows} decoder rows this pipeline was provisioned "
                                 "with (constructor argument num_beams / batch_size)")
            per = max(1, min(self.batch_size, self.max_rows // num_beams))
            if per < self.batch_size:
                _warn_once("rows", f"batch_size {self.batch_size} x {num_beams} beams exceeds {self.max_rows} decoder rows: "
                                   f"generating {per} chunks at a time")
        batches = [(n, mine[b0:b0 + per]) for n, b0 in enumerate(range(0, len(mine), per))]
        if len(self.engines) > 1 and len(batches) > 1:
            import concurrent.futures as cf
            # one worker per context; batch n always runs on context n % C, so a context is never re-entered
            lanes = [[b for b in batches if b[0] % len(self.engines) == c] for c in range(len(self.engines))]
            with cf.ThreadPoolExecutor(len(self.engines)) as ex:
                lane_out = list(ex.map(lambda lane: [run_batch(b) for b in lane], lanes))
            done = {b[0]: r for lane, outs in zip(lanes, lane_out) for b, r in zip(lane, outs)}
            results = [done[n] for n, _ in batches]
        else:
            results = [run_batch(b) for b in batches]
        recs = [r for rs, _ in results for r in rs]
        self.stats["generate_calls"] = self.stats.get("generate_calls", 0) + sum(c for _, c in results)
        recs = np.stack(recs) if recs else np.zeros((0, dist.REC_WORDS), np.int32)
        max_per_rank = max(h - l for l, h in dist.shard_bounds(len(windows), self.shard.world))
        t_ph.append(_time.perf_counter())
        allr = self.shard.all_gather_records(recs, max_per_rank)
        t_ph.append(_time.perf_counter())
        outputs = []
        for r in allr:
            _, toks, ts, stride = dist.unpack_record(r)
            o = {"tokens": toks, "token_timestamps": ts}
            if with_stride:
                o["stride"] = stride
            outputs.append(o)
        text, words = collate.decode_asr(self.vocab, outputs, time_precision=0.02, warn=logger.warning,
                                         return_timestamps="word" if rt == "word" else True)
        t_ph.append(_time.perf_counter())
        # where the wall time of the last call went on this rank (bench.py's long-form leg prints it: the scaling model of
        # DESIGN.md section 5 needs the rank-local part, which shrinks with the rank count, apart from the rest, which does not)
        self.stats["last_call_phase_s"] = {"load": t_ph[1] - t_ph[0], "local_batches": t_ph[2] - t_ph[1], "gather": t_ph[3] - t_ph[2],
                                           "collate_all_chunks": t_ph[4] - t_ph[3], "local_chunks": len(mine), "all_chunks": len(windows)}
        return {"text": text, "chunks": words}


def pipeline(task: str = "automatic-speech-recognition", model=None, tokenizer=None, feature_extractor=None, **kwargs):
    """Factory with the signature the reference uses (REF/transcribe.py:21-31)."""
    if task != "automatic-speech-recognition":
        raise KeyError(f"Unknown task {task}, available tasks are ['automatic-speech-recognition']")
    if model is None:
        raise ValueError("a model (WhisperForConditionalGeneration or ModelBundle) is required")
    return CrisperWhisperPipeline(model, tokenizer=tokenizer, feature_extractor=feature_extractor, **kwargs)
