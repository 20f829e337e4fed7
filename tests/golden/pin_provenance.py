"""Provenance of the PhiFlow pin fixtures (tests/golden/phiflow_*.npz): what make_phiflow_fixtures.py records inside each file and
what tests/test_phiflow_pin.py verifies before it lets a fixture pin anything.  Pure numpy / hashlib: runs under the reference's
python 3.6 / 3.7 as well as here.

A pin fixture is only as good as the link between its INPUTS and the committed oracle fixture the oracle is then run on, and
between its OUTPUTS and the reference's code.  Recorded per file:
    input_fixture            name of the committed oracle fixture whose inputs were fed to the reference (e.g. karman_step_64x32.npz)
    input_sha256             sha256 over the input arrays of that fixture (names, dtypes, shapes, bytes; input_keys() below)
    reference_commit         `git rev-parse HEAD` of the reference checkout the definitions were lifted from ("unknown" outside git)
    reference_scripts_sha256 sha256 of the reference script(s) the definitions came from (a hash -- no text of the reference is stored)
    generator_sha256         sha256 of make_phiflow_fixtures.py as it ran
    phiflow_version, tensorflow_version, python_version, created_utc
Verified: the input hash against the committed fixture (a fixture made from other inputs, or from a later edit of the inputs, is
refused), the stored copies of the inputs bit for bit, the presence of every field, phiflow 1.5.x / tensorflow 1.15.x."""
import hashlib

import numpy as np

INPUT_KEYS = {
    "karman_step": ("d", "vy", "vx", "re", "wy", "wx"),
    "burgers_step": ("vy", "vx", "fy", "fx", "wy", "wx", "dt", "nu"),
    "train": ("d", "vy", "vx", "re", "gt_vy", "gt_vx", "std_v", "std_re", "biases"),
}
FIELDS = ("input_fixture", "input_sha256", "reference_commit", "reference_scripts_sha256", "generator_sha256",
          "phiflow_version", "tensorflow_version", "python_version", "created_utc")


def kind_of(fixture_name):
    n = fixture_name.replace("phiflow_", "")
    return "karman_step" if n.startswith("karman_step") else ("burgers_step" if n.startswith("burgers_step") else "train")


def input_sha256(arrays, keys):
    """sha256 over (name, dtype, shape, C-order bytes) of arrays[k] for k in keys -- independent of the npz container and its compression"""
    h = hashlib.sha256()
    for k in keys:
        a = np.ascontiguousarray(np.asarray(arrays[k]))
        h.update(("%s|%s|%s|" % (k, a.dtype.str, ",".join(str(s) for s in a.shape))).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def verify(pin, committed_inputs, input_fixture_name):
    """Raises AssertionError (with the reason) unless `pin` (a loaded phiflow_*.npz) carries a complete provenance record that ties
    it to `committed_inputs` (the loaded committed oracle fixture `input_fixture_name`)."""
    missing = [f for f in FIELDS if f not in pin]
    assert not missing, "pin fixture lacks provenance fields %s: written by hand or by an old generator -- it pins nothing" % missing
    s = lambda k: str(np.asarray(pin[k]).item()) if np.asarray(pin[k]).shape == () else str(pin[k])
    assert s("input_fixture") == input_fixture_name, "pin fixture was made from %s, expected %s" % (s("input_fixture"), input_fixture_name)
    keys = INPUT_KEYS[kind_of(input_fixture_name)]
    want = input_sha256(committed_inputs, keys)
    assert s("input_sha256") == want, ("pin fixture's inputs (sha256 %s) are not the committed %s (sha256 %s): regenerate it "
                                       "(tests/golden/PIN.md)" % (s("input_sha256")[:16], input_fixture_name, want[:16]))
    if kind_of(input_fixture_name) != "train":          # (the train pin stores results only; its inputs live in train_16x8_sol2.npz)
        for k in keys:
            assert np.array_equal(np.asarray(pin[k]), np.asarray(committed_inputs[k])), "stored copy of input %r differs from %s" % (k, input_fixture_name)
    assert s("phiflow_version").startswith("1.5"), "generated with phiflow %s; the reference pins 1.5.1 (README.md:19-24)" % s("phiflow_version")
    assert s("tensorflow_version").startswith("1.15"), "generated with tensorflow %s; the reference pins 1.15" % s("tensorflow_version")
    for k in ("reference_scripts_sha256", "generator_sha256"):
        assert len(s(k)) == 64, "%s is not a sha256" % k
    assert len(s("reference_commit")) >= 7, "reference_commit missing"
    return {f: s(f) for f in FIELDS}
