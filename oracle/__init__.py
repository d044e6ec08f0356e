"""CPU oracle for the CrisperWhisper inference-and-alignment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  ``crisperwhisper_amd`` never imports this package and
fails loudly when its HIP library is missing.

What it restates
----------------
The reference (``/root/reference``, 258 lines of glue) delegates all arithmetic of the path
to the un-vendored, unpinned PyPI dependency ``transformers`` (``REF/requirements.txt:3``);
the version installed in the build image is **transformers 5.15.0**, referred to as ``TF/``
below (``/usr/local/lib/python3.10/dist-packages/transformers``).  Each module restates one
stage in numpy (plus one C file for the DTW inner loop) and cites the TF file:line it follows:

* ``mel.py``        TF/models/whisper/feature_extraction_whisper.py:135-168, TF/audio_utils.py:448-729
* ``model.py``      TF/models/whisper/modeling_whisper.py:215-505, 590-795, 1080
* ``logits.py``     TF/generation/logits_process.py:1816-2047 (+ MinNewTokens :203-260)
* ``generate.py``   TF/models/whisper/generation_whisper.py:383-968, 970-1116, 1977-2074;
                    TF/generation/utils.py:2783-2973 (greedy ``_sample``)
* ``timestamps.py`` TF/models/whisper/generation_whisper.py:43-115, 241-381
* ``dtw.c``         TF/models/whisper/generation_whisper.py:64-115 (C, for speed)
* ``collate.py``    TF/models/whisper/tokenization_whisper.py:901-1406
* ``pauses.py``     REF/utils.py:1-29
* ``pipeline.py``   TF/pipelines/automatic_speech_recognition.py:61-84, 345-710
* ``audio.py``      audio ingest in front of the path: torchaudio.functional.resample defaults (torchaudio is absent
                    offline: published algorithm restated, **parity unpinned** for this module only),
                    ffmpeg's f32le mono sample decoding, REF/app.py:85-93 normalisation

Parity pinning
--------------
The reference ships no tests, golden vectors or fixtures for this path.
The oracle is therefore pinned against *outputs of the reference's own dependency run in
the build container*: ``tests/golden/gen_golden.py`` imports transformers 5.15.0, runs
each stage / the whole ``pipeline(..., return_timestamps="word")`` call of
``REF/transcribe.py:21-33`` on seeded synthetic inputs and commits the results under
``tests/golden/*.npz|json``; ``tests/test_oracle_vs_golden.py`` checks every oracle stage
against them (and live against transformers when it is importable).
"""
