"""Build-container-only cross-check: everything the oracle writes for which no C reference
exists (1-state streams, PICS, MIC2 incl. temporal, MIC3) must decode to the original samples
with the reference's independent JavaScript decoder (web/mic-decoder.js).  Skipped where
/root/reference or node is absent (e.g. on the GPU box)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_JS = "/root/reference/web/mic-decoder.js"
pytestmark = pytest.mark.skipif(not (os.path.exists(REF_JS) and shutil.which("node")), reason="reference JS decoder / node not available")


def _run(tmp_path, jobs):
    dec = tmp_path / "mic-decoder.mjs"            # node 12 needs the .mjs suffix for ESM; tmp copy, never committed
    shutil.copy(REF_JS, dec)
    jp = tmp_path / "job.json"
    jp.write_text(json.dumps(jobs))
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js_crosscheck.mjs"), str(dec), str(jp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_js_decoder_accepts_oracle_streams(mico, synth, tmp_path):
    mr = np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    mx = int(mr.max())
    jobs, want = [], []

    def add(kind, blob, arr, **kw):
        i = len(jobs)
        p_in, p_out = tmp_path / f"in{i}.bin", tmp_path / f"out{i}.bin"
        p_in.write_bytes(blob)
        jobs.append(dict(kind=kind, **{"in": str(p_in)}, out=str(p_out), **kw))
        want.append((p_out, arr))

    tok = mico.delta_rle_compress(mr, mx)
    for ns in (1, 2, 4, 8):                                   # 1-state has no C reference
        rc, blob = mico.fse_compress(tok, ns)
        assert rc == 0
        add("frame", blob, mr, width=256, height=256)
    for strips, ns in ((1, 2), (8, 2), (8, 4), (5, 8)):
        rc, blob = mico.pics_compress(mr, mx, strips, ns)
        assert rc == 0
        add("file", blob, mr)
    stack = synth.ct_stack(frames=4, size=128, depth=12, seed=3)
    for temporal in (False, True):
        rc, blob = mico.mic2_compress(stack, 4095, temporal)
        assert rc == 0
        add("mic2", blob, stack)
    img = synth.wsi_like(300, 256, seed=9)
    rc, blob = mico.wsi_compress(img)
    assert rc == 0
    add("file", blob, img)
    rc, blob = mico.micr_write(img[:200, :280])               # MICR: header + CompressRGB blob
    assert rc == 0
    add("file", blob, np.ascontiguousarray(img[:200, :280]))
    for ns in (2, 4, 8):                                      # MIC1: header + CompressSingleFrame{,4State,8State}
        rc, blob = mico.mic1_write(mr, mx, ns)
        assert rc == 0
        add("file", blob, mr)
    _run(tmp_path, jobs)
    for p_out, arr in want:
        got = np.frombuffer(p_out.read_bytes(), dtype=arr.dtype)
        assert got.size == arr.size and np.array_equal(got, arr.reshape(-1)), p_out.name
