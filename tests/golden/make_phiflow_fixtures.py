#!/usr/bin/env python
"""PIN KIT -- turns "parity unpinned" into a one-command job for anyone who HAS the reference's dependencies.

The arithmetic of the hot path lives in phiflow==1.5.1 (commit 4f5e678) and TensorFlow 1.15 (/root/reference/README.md:19-24);
neither is installable in the build container and the reference ships no tests or golden vectors, so
every fixture in this directory was produced by oracle/sol_oracle.py itself.  This script produces fixtures from the REFERENCE:
it loads the inputs of three committed oracle fixtures, runs the reference's OWN code on them and writes

    tests/golden/phiflow_karman_step_64x32.npz    KarmanFlow.step                 karman-2d/karman_train.py:166-185
    tests/golden/phiflow_karman_step_16x8.npz     (same, the small grid)
    tests/golden/phiflow_burgers_step_32x32.npz   BurgersTest.step_with_f         burgers/burgers_train.py:172-187
    tests/golden/phiflow_train_16x8_sol2.npz      the unrolled SOL-2 loss + grads karman-2d/karman_train.py:77-90,101-138,397-447

with the same array names as the oracle fixtures.  tests/test_phiflow_pin.py then compares the oracle with them (it skips while
they are absent) and reports which setting of the recalled choices Q2-Q7 reproduces PhiFlow.

    pip install phiflow==1.5.1 tensorflow==1.15      # python 3.6 / 3.7      (the whole recipe, expected sizes and what to commit: PIN.md)
    python tests/golden/make_phiflow_fixtures.py --reference /path/to/Solver-in-the-Loop

Every file carries a provenance record (pin_provenance.py: sha256 of the committed inputs it was made from, `git rev-parse HEAD` and a
hash of the reference scripts, a hash of this generator, the package versions); tests/test_phiflow_pin.py verifies it, so a stale or
hand-made fixture cannot pin anything.

The reference's classes and functions are NOT restated here: `reference_defs` parses the reference's scripts and executes
exactly the `def` / `class` statements named (KarmanFlow, to_feature, to_staggered, model_mars_moon, BurgersTest,
BurgersVelocitySMAC) inside a namespace that holds `from phi.tf.flow import *` -- the scripts themselves cannot be imported
(they parse argv and train at import time).  Only the driver lines around them (placeholders, the msteps loop, the loss) follow
this file, each with the reference line it mirrors.  Nothing of this travels to the GPU box; the outputs are data.

STATUS: written against the PhiFlow 1.5.1 API as the reference uses it, NOT executed in the build container (no phi / tf there).
If an attribute differs in your installation the failure is an AttributeError in the few driver lines below, not a silent
mismatch."""
import argparse
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pin_provenance as prov  # noqa: E402


def provenance(ref, input_fixture, scripts, versions):
    """The record tests/test_phiflow_pin.py verifies (tests/golden/pin_provenance.py): which inputs, which reference, which generator."""
    import datetime
    import hashlib
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", ref, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                universal_newlines=True).stdout.strip() or "unknown"
    except OSError:
        commit = "unknown"
    h = hashlib.sha256()
    for rel_path in scripts:                       # a HASH of the reference scripts the definitions were lifted from; no text is stored
        h.update(rel_path.encode())
        h.update(prov.file_sha256(os.path.join(ref, rel_path)).encode())
    z = np.load(os.path.join(HERE, input_fixture))
    return {"input_fixture": input_fixture, "input_sha256": prov.input_sha256(z, prov.INPUT_KEYS[prov.kind_of(input_fixture)]),
            "reference_commit": commit if len(commit) >= 7 else "unknown", "reference_scripts_sha256": h.hexdigest(),
            "generator_sha256": prov.file_sha256(os.path.abspath(__file__)),
            "phiflow_version": versions[0], "tensorflow_version": versions[1], "python_version": sys.version.split()[0],
            "created_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ")}


def need_reference_stack():
    try:
        import phi  # noqa: F401
        import tensorflow as tf  # noqa: F401
    except ImportError as e:
        raise SystemExit("make_phiflow_fixtures.py needs the reference's own dependencies (phiflow==1.5.1 @4f5e678 and tensorflow 1.15, "
                         "/root/reference/README.md:19-24): %s.\nThey are not installable in the build container; run this where they are, then "
                         "commit tests/golden/phiflow_*.npz." % e)
    import phi
    import tensorflow as tf
    ver = getattr(phi, "__version__", "?")
    if not str(ver).startswith("1.5"):
        print("WARNING: phiflow %s found, the reference pins 1.5.1 -- the fixtures will pin THAT version" % ver, file=sys.stderr)
    return str(ver), tf.__version__


def reference_defs(path, names, namespace):
    """Execute the top-level `def` / `class` statements called `names` of the reference script `path` inside `namespace`."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    found = {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            found[node.name] = node
    missing = [n for n in names if n not in found]
    if missing:
        raise SystemExit("%s does not define %s" % (path, missing))
    for n in names:                                   # in the order asked for (base classes first)
        mod = ast.Module(body=[found[n]], type_ignores=[])
        exec(compile(mod, path, "exec"), namespace)
    return namespace


def phi_tf_namespace():
    ns = {}
    exec("from phi.tf.flow import *\nimport phi.tf.util\nimport tensorflow as tf\nfrom tensorflow import keras\nimport numpy as np", ns)
    return ns


def _session(tf):
    cfg = tf.compat.v1.ConfigProto(device_count={"GPU": 0})            # CPU: the fixtures pin arithmetic, not speed
    return tf.compat.v1.Session(config=cfg)


def _c(a):
    """[B,H,W] -> [B,H,W,1] float32 (PhiFlow field data carries a channel axis)"""
    return np.asarray(a, dtype=np.float32)[..., None]


def velocity_bc(Y, X, B):
    """velBCy / velBCyMask of karman-2d/karman_train.py:366-373 (v_y component, shape [B,Y+1,X,1])"""
    vn = np.zeros((B, Y + 1, X, 1))
    vn[..., 0:2, 0:vn.shape[2] - 1, 0] = 1.0
    vn[..., 0:vn.shape[1], 0:1, 0] = 1.0
    vn[..., 0:vn.shape[1], -1:, 0] = 1.0
    return vn


def karman_fluid(ns, B, Y, X):
    """st_co of karman_train.py:363: Fluid on Domain([Y, X], box[0:200, 0:100], OPEN), buoyancy_factor = 0"""
    return ns["Fluid"](ns["Domain"](resolution=[Y, X], box=ns["box"][0:200, 0:100], boundaries=ns["OPEN"]), buoyancy_factor=0, batch_size=B)


def karman_step_fixture(ref, name):
    ns = reference_defs(os.path.join(ref, "karman-2d", "karman_train.py"), ["KarmanFlow"], phi_tf_namespace())
    tf = ns["tf"]
    z = np.load(os.path.join(HERE, name + ".npz"))
    B, Y, X = z["d"].shape
    st = karman_fluid(ns, B, Y, X)
    st_in = ns["phi"].tf.util.placeholder_like(st)                        # karman_train.py:376
    re_in = tf.compat.v1.placeholder(tf.float32, shape=[B])              # :378
    vn = velocity_bc(Y, X, B)
    out = ns["KarmanFlow"]().step(st_in, re=re_in, res=X, velBCy=vn, velBCyMask=vn)      # :403-409 (res = Rlo[-1])
    vy_in, vx_in = st_in.velocity.data[0].data, st_in.velocity.data[1].data
    vy_out, vx_out = out.velocity.data[0].data, out.velocity.data[1].data
    obj = tf.reduce_sum(vy_out * _c(z["wy"])) + tf.reduce_sum(vx_out * _c(z["wx"]))
    gy, gx = tf.gradients(obj, [vy_in, vx_in])
    with _session(tf) as sess:
        r = sess.run([out.density.data, vy_out, vx_out, gy, gx],
                     feed_dict={st_in.density.data: _c(z["d"]), vy_in: _c(z["vy"]), vx_in: _c(z["vx"]), re_in: z["re"].astype(np.float32)})
    s = lambda a: np.asarray(a)[..., 0]
    return dict(d=z["d"], vy=z["vy"], vx=z["vx"], re=z["re"], wy=z["wy"], wx=z["wx"],
                d_out=s(r[0]), vy_out=s(r[1]), vx_out=s(r[2]), g_vy=s(r[3]), g_vx=s(r[4]))


def burgers_step_fixture(ref):
    ns = reference_defs(os.path.join(ref, "burgers", "burgers_train.py"), ["BurgersVelocitySMAC", "BurgersTest"], phi_tf_namespace())
    tf = ns["tf"]
    z = np.load(os.path.join(HERE, "burgers_step_32x32.npz"))
    B, Y, X = z["vy"].shape[0], z["vy"].shape[1] - 1, z["vy"].shape[2]
    dm = ns["Domain"](resolution=[Y, X], box=ns["box"]([Y, X]), boundaries=ns["PERIODIC"])         # burgers_train.py:346 (len = res, Makefile:71)
    st = ns["BurgersVelocitySMAC"](dm, batch_size=B)                                                  # :348
    v_in = ns["phi"].tf.util.placeholder_like(st)
    f_in = ns["phi"].tf.util.placeholder_like(st)
    sim = ns["BurgersTest"](default_viscosity=float(z["nu"]))
    out = sim.step_with_f(v=v_in, f=f_in, dt=float(z["dt"]))                                          # :393-398
    comp = lambda s_: (s_.velocity.data[0].data, s_.velocity.data[1].data)
    (vy_in, vx_in), (fy_in, fx_in), (vy_out, vx_out) = comp(v_in), comp(f_in), comp(out)
    obj = tf.reduce_sum(vy_out * _c(z["wy"])) + tf.reduce_sum(vx_out * _c(z["wx"]))
    gy, gx = tf.gradients(obj, [vy_in, vx_in])
    with _session(tf) as sess:
        r = sess.run([vy_out, vx_out, gy, gx], feed_dict={vy_in: _c(z["vy"]), vx_in: _c(z["vx"]), fy_in: _c(z["fy"]), fx_in: _c(z["fx"])})
    s = lambda a: np.asarray(a)[..., 0]
    return dict(vy=z["vy"], vx=z["vx"], fy=z["fy"], fx=z["fx"], wy=z["wy"], wx=z["wx"], dt=z["dt"], nu=z["nu"],
                vy_out=s(r[0]), vx_out=s(r[1]), g_vy=s(r[2]), g_vx=s(r[3]))


def train_fixture(ref):
    """The unrolled SOL-2 graph of karman_train.py:392-447 on the inputs of train_16x8_sol2.npz: loss, per-step losses, final
    state, gradient w.r.t. the 24 Keras variables (stored like the oracle fixture: per-tensor norms + every 16th element)."""
    sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
    import torch
    import sol_oracle as o                                       # ONLY for the seeded Glorot weights (inputs, not results)
    ns = reference_defs(os.path.join(ref, "karman-2d", "karman_train.py"), ["to_feature", "to_staggered", "model_mars_moon", "KarmanFlow"], phi_tf_namespace())
    tf, keras = ns["tf"], ns["keras"]
    z = np.load(os.path.join(HERE, "train_16x8_sol2.npz"))
    B, Y, X = z["d"].shape
    ms = z["gt_vy"].shape[0]
    std_v, std_re = tuple(float(v) for v in z["std_v"]), float(z["std_re"])
    st = karman_fluid(ns, B, Y, X)
    st_in = ns["phi"].tf.util.placeholder_like(st)
    re_in = tf.compat.v1.placeholder(tf.float32, shape=[B])
    gt_in = [ns["phi"].tf.util.placeholder_like(st) for _ in range(ms)]
    vn = velocity_bc(Y, X, B)
    sim = ns["KarmanFlow"]()
    sess = _session(tf)
    tf.compat.v1.keras.backend.set_session(sess)                                                      # :390
    model = ns["model_mars_moon"](ns["to_feature"](st_in, re_in))                                    # :394-395
    prd = []
    for i in range(ms):                                                                                # :399-426
        s_ = sim.step(st_in if i == 0 else prd[-1], re=re_in, res=X, velBCy=vn, velBCyMask=vn)
        corr = ns["to_staggered"](model(ns["to_feature"](s_, re_in) / [std_v[0], std_v[1], std_re]) * std_v, box=st.velocity.box)
        prd.append(s_.copied_with(velocity=s_.velocity + corr))
    loss_steps = [tf.nn.l2_loss((gt_in[i].velocity.staggered_tensor() - prd[i].velocity.staggered_tensor()) / std_v) for i in range(ms)]   # :428-435
    loss = tf.reduce_sum(loss_steps) / ms                                                              # :436
    sess.run(tf.compat.v1.global_variables_initializer())
    params = [p.detach().float().numpy() for p in o.init_params(0)]                                   # the oracle fixture's weights: seed 0 ...
    off = 0
    for k, p in enumerate(params):                                                                     # ... with the stored biases
        if p.ndim == 1:
            params[k] = z["biases"][off:off + p.size].astype(np.float32)
            off += p.size
    model.set_weights(params)                                                                          # Keras order = [kernel, bias] x 12
    grads = tf.gradients(loss, model.trainable_weights)
    comp = lambda s_: (s_.velocity.data[0].data, s_.velocity.data[1].data)
    feed = {st_in.density.data: _c(z["d"]), comp(st_in)[0]: _c(z["vy"]), comp(st_in)[1]: _c(z["vx"]), re_in: z["re"].astype(np.float32)}
    for i in range(ms):
        feed[comp(gt_in[i])[0]] = _c(z["gt_vy"][i])
        feed[comp(gt_in[i])[1]] = _c(z["gt_vx"][i])
        feed[gt_in[i].density.data] = _c(np.zeros_like(z["d"]))
    r = sess.run([loss, loss_steps, prd[-1].density.data, comp(prd[-1])[0], comp(prd[-1])[1], grads], feed_dict=feed)
    sess.close()
    flat = np.concatenate([np.asarray(g).ravel() for g in r[5]])
    s = lambda a: np.asarray(a)[..., 0]
    return dict(loss=float(r[0]), loss_steps=np.asarray(r[1], dtype=np.float64), d_final=s(r[2]), vy_final=s(r[3]), vx_final=s(r[4]),
                grad_norms=np.array([float(np.linalg.norm(np.asarray(g).ravel())) for g in r[5]]), grads_sub16=flat[::16].astype(np.float32))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="/root/reference", help="checkout of tum-pbs/Solver-in-the-Loop")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None, help="comma separated subset of: step64,step16,burgers,train")
    args = ap.parse_args()
    versions = need_reference_stack()
    if not os.path.exists(os.path.join(args.reference, "karman-2d", "karman_train.py")):
        raise SystemExit("--reference %s: karman-2d/karman_train.py not found" % args.reference)
    KT, BT = os.path.join("karman-2d", "karman_train.py"), os.path.join("burgers", "burgers_train.py")
    # key -> (output file, input fixture, reference scripts the definitions come from, job)
    jobs = {"step64": ("phiflow_karman_step_64x32.npz", "karman_step_64x32.npz", [KT], lambda: karman_step_fixture(args.reference, "karman_step_64x32")),
            "step16": ("phiflow_karman_step_16x8.npz", "karman_step_16x8.npz", [KT], lambda: karman_step_fixture(args.reference, "karman_step_16x8")),
            "burgers": ("phiflow_burgers_step_32x32.npz", "burgers_step_32x32.npz", [BT], lambda: burgers_step_fixture(args.reference)),
            "train": ("phiflow_train_16x8_sol2.npz", "train_16x8_sol2.npz", [KT], lambda: train_fixture(args.reference))}
    for key in (args.only.split(",") if args.only else jobs):
        fname, infix, scripts, fn = jobs[key]
        data = fn()
        data.update(provenance(args.reference, infix, scripts, versions))       # verified by tests/test_phiflow_pin.py before the file may pin anything
        np.savez_compressed(os.path.join(args.out, fname), **data)
        print("wrote", os.path.join(args.out, fname), os.path.getsize(os.path.join(args.out, fname)) // 1024, "KB")


if __name__ == "__main__":
    main()
