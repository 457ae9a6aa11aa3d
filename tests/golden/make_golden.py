"""Regenerates tests/golden/stream_md5.json: MD5 of each synthetic stream of the parity matrix and of
its decoded frames (oracle output, which equals the generator's own reconstruction).  The vectors
pin generator + oracle against silent behavioural drift; they do not come from the reference, which
has no test vectors."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle  # noqa: E402
import streamgen  # noqa: E402
from conftest import FIELD_CABAC_MATRIX, FIELD_MATRIX, MATRIX, POC_MATRIX  # noqa: E402

out = {}
for name in sorted(MATRIX):
    s, rec, _ = streamgen.encode(**MATRIX[name])
    frames, _ = oracle.decode(s, crop=False)
    assert (frames == rec).all(), name
    out[name] = {"stream_md5": hashlib.md5(s).hexdigest(), "frames_md5": hashlib.md5(frames.tobytes()).hexdigest(), "stream_bytes": len(s)}
json.dump(out, open(os.path.join(HERE, "stream_md5.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "vectors")

# field pictures (oracle + generator only): their own file, so that the GPU golden test keeps reading stream_md5.json whole
out = {}
for name in sorted(FIELD_MATRIX):
    s, rec, _ = streamgen.encode(**FIELD_MATRIX[name])
    frames, _ = oracle.decode(s, crop=False)
    assert (frames == rec).all(), name
    out[name] = {"stream_md5": hashlib.md5(s).hexdigest(), "frames_md5": hashlib.md5(frames.tobytes()).hexdigest(), "stream_bytes": len(s)}
json.dump(out, open(os.path.join(HERE, "field_md5.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "field vectors")

# the same recipes with CABAC (unpinned context values of field-coded blocks: these vectors pin the mechanism and guard against drift, nothing more)
out = {}
for name in sorted(FIELD_CABAC_MATRIX):
    s, rec, _ = streamgen.encode(**FIELD_CABAC_MATRIX[name])
    frames, _ = oracle.decode(s, crop=False)
    assert (frames == rec).all(), name
    out[name] = {"stream_md5": hashlib.md5(s).hexdigest(), "frames_md5": hashlib.md5(frames.tobytes()).hexdigest(), "stream_bytes": len(s)}
json.dump(out, open(os.path.join(HERE, "field_cabac_md5.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "CABAC field vectors")

# separate bottom-field picture order counts (CPU-checked cases: oracle, generator, and the product's host side against the null device)
out = {}
for name in sorted(POC_MATRIX):
    s, rec, _ = streamgen.encode(**POC_MATRIX[name])
    frames, _ = oracle.decode(s, crop=False)
    assert (frames == rec).all(), name
    out[name] = {"stream_md5": hashlib.md5(s).hexdigest(), "frames_md5": hashlib.md5(frames.tobytes()).hexdigest(), "stream_bytes": len(s),
                 "pocs": [int(x) for x in streamgen.last_pocs()]}
json.dump(out, open(os.path.join(HERE, "poc_md5.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "POC vectors")
