"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the
ctypes structs match the header, the field recogniser flattens both our combinator objects and the
closure tree an unchanged scripts/main.py builds, and nothing in the product imports the oracle."""
import ctypes
import functools
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from vsrd_amd import _lib
    return _lib.load()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "vsrd_hip.h")).read()
    declared = set(re.findall(r"^(?:int32_t|size_t|const char\*)\s+(vsrd_\w+)\s*\(", header, flags=re.M))
    assert {"vsrd_render_forward", "vsrd_render_backward", "vsrd_render_hierarchical_forward", "vsrd_field_eval",
            "vsrd_ray_directions", "vsrd_sample_stratified", "vsrd_sample_importance"} <= declared
    from vsrd_amd import _lib
    assert declared == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    for name in declared:
        assert getattr(lib, name) is not None
    # ... and the converse: the library exports no vsrd_* symbol the header does not declare (nm -D; the debug build's phase clocks aside)
    import subprocess
    exported = {line.split()[-1] for line in subprocess.run(["nm", "-D", "--defined-only", _lib.LIBRARY_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
                if line.split()[-1].startswith("vsrd_") and " T " in line}
    assert exported - {"vsrd_debug_phase_cycles"} == declared, exported ^ declared
    assert lib.vsrd_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define VSRD_ABI_VERSION (\d+)", header).group(1))
    assert lib.vsrd_workspace_bytes(16, 0) == 16384 * 4 * 16 * 16 * 4
    # box partials + per-wave MLP partials (512 workgroups x 2 waves) + residual jets [wave][4 rounds][N][64] float4 + seeds [wave][8 rays][4][N][10][64]
    assert lib.vsrd_workspace_bytes(16, 1) == 16384 * 4 * 16 * 16 * 4 + 512 * 2 * 16 * 1617 * 4 + 512 * 2 * 4 * 16 * 64 * (16 + 8 * 28) + 512 * 2 * 16 * 16
    assert lib.vsrd_workspace_bytes(0, 0) == 0 and lib.vsrd_workspace_bytes(65, 0) == 0
    assert lib.vsrd_error_string(-1) == b"invalid argument"


def test_struct_layout_matches_header(tmp_path):
    """The ctypes structures against the C compiler's view of include/vsrd_hip.h itself (gcc: sizeof / offsetof of every member that
    matters), not against numbers copied into the test."""
    import subprocess
    from vsrd_amd import _lib
    members = {"vsrd_field": (_lib.Field, ["instances", "mlp_weights"]),
               "vsrd_render_config": (_lib.RenderConfig, ["seed", "stream_offset", "flags", "device_schedule", "device_stream_offset", "ray_indices", "rays_per_origin",
                                                          "target_columns", "target_stride", "out_distances", "out_u_fine", "num_frames", "frame_stride", "adjoint_slots_per_item"]),
               "vsrd_frame_config": (_lib.FrameConfig, ["num_steps", "lr_gamma", "num_frames", "frame_stride"]),
               "vsrd_adam_tensors": (_lib.AdamTensors, ["learning_rate"]),
               "vsrd_hypernetwork": (_lib.Hypernetwork, ["embeddings", "norm_bias", "num_frames", "frame_stride"])}
    lines = []
    for struct, (_, names) in members.items():
        lines.append(f'printf("{struct} %zu\\n", sizeof({struct}));')
        lines += [f'printf("{struct}.{name} %zu\\n", offsetof({struct}, {name}));' for name in names]
    source = tmp_path / "layout.c"
    source.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vsrd_hip.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0;\n}\n")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(source), "-o", str(tmp_path / "layout")], check=True)
    seen = dict(line.split() for line in subprocess.run([str(tmp_path / "layout")], capture_output=True, text=True, check=True).stdout.splitlines())
    for struct, (binding, names) in members.items():
        assert int(seen[struct]) == ctypes.sizeof(binding), struct
        for name in names:
            assert int(seen[f"{struct}.{name}"]) == getattr(binding, name).offset, (struct, name)
    assert ctypes.sizeof(_lib.RenderConfig) == 160 and ctypes.sizeof(_lib.FrameConfig) == 128 and ctypes.sizeof(_lib.Hypernetwork) == 24 + 24 * 40 + 16      # ABI 8


def test_cpu_tensors_are_rejected_not_emulated(lib):
    from vsrd_amd import _lib
    with pytest.raises(_lib.VsrdHipError):
        _lib.ptr(torch.zeros(3))


def _main_py_style_field(loc, dim, rot, temperature, mlp=None, hyper=None):
    """Closures written like scripts/main.py:433-509 (nested ``wrapper`` functions), around OUR sdfs objects."""
    from vsrd_amd import rendering

    def residual_distance_field(distance_field):
        def wrapper(positions):
            return torch.sigmoid(distance_field(positions) - 1.0)     # never evaluated: the recogniser reads the closure
        return wrapper

    def residual_composition(distance_field, residual_distance_field):
        def wrapper(positions):
            return distance_field(positions) + residual_distance_field(positions)
        return wrapper

    def instance_field(distance_field, instance_label):
        def wrapper(positions):
            return distance_field(positions), instance_label
        return wrapper

    def soft_union(distance_fields, temperature):
        def wrapper(positions):
            return [f(positions) for f in distance_fields], temperature
        return wrapper

    members = []
    for i in range(loc.shape[0]):
        base = rendering.sdfs.box(dim[i])
        if mlp is not None:
            base = residual_composition(distance_field=base, residual_distance_field=residual_distance_field(
                distance_field=functools.partial(hyper, mlp[i])))
        members.append(rendering.sdfs.translation(rendering.sdfs.rotation(
            instance_field(distance_field=base, instance_label=dim.new_tensor(i, dtype=torch.long)), rot[i]), loc[i]))
    return soft_union(distance_fields=members, temperature=temperature)


def test_hypernetwork_pointer_block_follows_the_module_and_the_optimiser(lib):
    """optimization.hypernetwork_tensors: the vsrd_hypernetwork block points at the torch module's own parameters and at
    torch.optim.Adam's own state (created on the spot), group by group; other architectures are refused, and the entry points reject
    what they cannot run (no compute here: there is no GPU)."""
    from vsrd_amd import _lib, models, optimization
    net = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
    embeddings = torch.nn.Parameter(torch.randn(1, 5, 256))
    lr = lambda v: torch.tensor(v)
    optimiser = torch.optim.Adam([dict(params=[embeddings], lr=lr(1e-3)), dict(params=list(net.parameters()), lr=lr(1e-4))], lr=lr(1e-3))
    block = optimization.hypernetwork_tensors(net, embeddings, optimiser, 0.99)
    assert (block.num_instances, block.num_outputs) == (5, _lib.MLP_WEIGHTS) and abs(block.lr_gamma - 0.99) < 1e-7
    assert (block.beta1, block.beta2) == (ctypes.c_float(0.9).value, ctypes.c_float(0.999).value)
    assert block.embeddings.parameter == embeddings.data_ptr() and block.embeddings.learning_rate == optimiser.param_groups[0]["lr"].data_ptr()
    linears = [b[0] for b in net.hypernetwork]
    for l, linear in enumerate(linears):
        state = optimiser.state[linear.weight_v]
        assert block.weight_v[l].parameter == linear.weight_v.data_ptr() and block.weight_g[l].parameter == linear.weight_g.data_ptr()
        assert block.weight_v[l].exp_avg == state["exp_avg"].data_ptr() and block.weight_v[l].step == state["step"].data_ptr()
        assert block.bias[l].learning_rate == optimiser.param_groups[1]["lr"].data_ptr()
    assert block.norm_weight[3].parameter == net.hypernetwork[3][1].weight.data_ptr()
    assert len(optimiser.state) == 1 + 5 * 3 + 4 * 2 and all(float(s["step"]) == 0.0 for s in optimiser.state.values())
    with pytest.raises(ValueError):                                   # not the reference's hypernetwork
        optimization.hypernetwork_tensors(models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256]), embeddings, optimiser, 0.99)
    assert lib.vsrd_hypernetwork_workspace_bytes(0) == 0 and lib.vsrd_hypernetwork_workspace_bytes(65) == 0
    assert lib.vsrd_hypernetwork_workspace_bytes(8) > 8 * 256 * 4 * 100
    assert lib.vsrd_hypernetwork_forward(block, None, 0, None, None, None) == -1           # VSRD_E_INVALID_ARGUMENT: no output buffer
    block.num_outputs = 100
    assert lib.vsrd_hypernetwork_backward_step(block, None, 0, None, 1.0, None) == -1


def test_recogniser_flattens_main_py_closures():
    from vsrd_amd import fields
    g = torch.Generator().manual_seed(0)
    loc = torch.randn(5, 3, generator=g).requires_grad_(True)
    dim = (torch.rand(5, 3, generator=g) + 0.5).requires_grad_(True)
    rot = torch.randn(5, 3, 3, generator=g).requires_grad_(True)
    block = fields.flatten(_main_py_style_field(loc, dim, rot, 0.37))
    assert block.instances.shape == (5, 16) and block.temperature == pytest.approx(0.37)
    assert block.mlp_weights is None and block.label_indices is None and not block.hard
    assert torch.equal(block.instances[:, 0:3], loc) and torch.equal(block.instances[:, 3:12], rot.reshape(5, 9))
    assert torch.equal(block.instances[:, 12:15], dim) and torch.all(block.instances[:, 15] == 0)
    # autograd reaches the original tensors through the packing
    gl, gd, gr = torch.autograd.grad(block.instances.sum(), [loc, dim, rot])
    assert torch.all(gl == 1) and torch.all(gd == 1) and torch.all(gr == 1)
    # residual variant: weights travel through functools.partial (main.py:541-544)
    mlp = torch.randn(5, 1617, generator=g)
    block = fields.flatten(_main_py_style_field(loc, dim, rot, 1.0, mlp=mlp, hyper=lambda w, x: x))
    assert torch.equal(block.mlp_weights, mlp)


def test_block_hand_over_is_scoped_to_one_wrapper_call():
    """fields.BlockHandOver (the module-level "last block" cache of round 3 could return a stale block): the block of
    main.py's pass 1 serves pass 2 only for the same closure object with every captured tensor at the same version; flatten()
    itself caches nothing, so a parameter update between steps -- through an optimiser, `.data` or a raw pointer -- is always seen."""
    from vsrd_amd import fields
    g = torch.Generator().manual_seed(1)
    loc = torch.randn(4, 3, generator=g).requires_grad_(True)
    dim = (torch.rand(4, 3, generator=g) + 0.5).requires_grad_(True)
    rot = torch.randn(4, 3, 3, generator=g).requires_grad_(True)
    field = _main_py_style_field(loc, dim, rot, 0.5)
    with torch.no_grad():
        block = fields.flatten(field)
    assert block.instances.requires_grad                          # built with autograd on even under no_grad: pass 2 differentiates it
    assert block.capture_key is not None                          # flatten's walk of the tree also serves the hand-over's comparison
    hand = fields.BlockHandOver(field, block)
    assert hand.key == block.capture_key and hand.take(field) is block
    assert hand.take(field) is None                               # single use: the block carries ONE autograd graph
    assert fields.BlockHandOver(field, block).take(_main_py_style_field(loc, dim, rot, 0.5)) is None      # another closure object (the next step's): no reuse
    hand = fields.BlockHandOver(field, block)
    with torch.no_grad():
        loc.add_(1.0)                                             # an in-place update bumps the version
    assert hand.take(field) is None
    # flatten() has no memory: the step after a `.data` update (no version bump) sees the new values
    loc.data.add_(5.0)
    assert torch.equal(fields.flatten(field).instances[:, 0:3], loc)
    fields.BLOCK_HAND_OVER = False
    try:
        assert fields.BlockHandOver(field, block).take(field) is None
    finally:
        fields.BLOCK_HAND_OVER = True


def test_recogniser_objects_labels_and_rejections():
    from vsrd_amd import fields, rendering
    loc, dim, rot = torch.zeros(3, 3), torch.ones(3, 3), torch.eye(3).repeat(3, 1, 1)
    members = [rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(dim[i]), lab, 3), rot[i]), loc[i])
               for i, lab in enumerate([2, 0, 1])]
    block = fields.flatten(fields.soft_union(members, 0.5))
    assert block.label_indices.tolist() == [2, 0, 1]
    assert fields.flatten(fields.hard_union(members)).hard
    assert fields.flatten(rendering.sdfs.box(dim[0])).instances.shape == (1, 16)       # identity pose filled in
    with pytest.raises(fields.UnsupportedFieldError):
        fields.flatten(lambda p: p)
    with pytest.raises(fields.UnsupportedFieldError):                                   # rotation(translation(.)) is a different map
        fields.flatten(fields.soft_union([rendering.sdfs.rotation(rendering.sdfs.translation(rendering.sdfs.box(dim[0]), loc[0]), rot[0])], 1.0))
    with pytest.raises(fields.UnsupportedFieldError):
        fields.flatten(fields.soft_union([rendering.sdfs.box(dim[0])] * 65, 1.0))


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "vsrd_amd")):
        for name in files:
            if name.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(base, name)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{name} imports the oracle"
                assert "/root/reference" not in text


def test_formats_match_reference_writers():
    """N4: KITTI label lines byte-for-byte against the reference's convert_predictions.save_prediction (golden g12), the
    prediction JSON layout, the checkpoint's detector state-dict layout and the multi-view confidence voting."""
    import json
    import tempfile
    import numpy as np
    from conftest import load_golden
    from vsrd_amd import formats, models
    g = load_golden("g12_formats")
    want = bytes(g["kitti_text"].numpy().astype(np.uint8)).decode()
    got = "".join(formats.kitti_label_line("car", b3, b2, s) for b3, b2, s in zip(g["boxes_3d"], g["boxes_2d"], g["scores"]))
    for a, b in zip(got.splitlines(), want.splitlines()):
        fa, fb = a.split(" "), b.split(" ")
        assert fa[0] == fb[0] and len(fa) == len(fb) == 16
        np.testing.assert_allclose([float(x) for x in fa[1:]], [float(x) for x in fb[1:]], rtol=1e-5, atol=1e-5)
    assert got.count("\n") == want.count("\n") == 3
    keys = bytes(g["state_keys"].numpy().astype(np.uint8)).decode().split("\n")
    mine = [f"{k}:{tuple(v.shape)}" for k, v in models.BoxParameters3D(1, 3).state_dict().items()]
    assert mine == keys                                           # loadable by make_predictions.py:61-66
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "predictions", "frame.json")
        formats.save_prediction(path, g["boxes_3d"], g["boxes_2d"], g["scores"])
        record = json.load(open(path))
        assert list(record) == ["boxes_3d", "boxes_2d", "confidences"] and list(record["boxes_3d"]) == ["car"]
        assert np.asarray(record["boxes_3d"]["car"]).shape == (3, 8, 3) and np.asarray(record["boxes_2d"]["car"]).shape == (3, 2, 2)
    # voting: prediction 0 overlaps target instance 1 in both views, prediction 1 overlaps instance 0 in one view
    pd = [torch.tensor([[[0., 0.], [10., 10.]], [[20., 20.], [30., 30.]]])] * 2
    gt = [torch.tensor([[[20., 20.], [30., 30.]], [[0., 0.], [10., 10.]]]), torch.tensor([[[0., 0.], [10., 5.]]])]
    conf, pd_idx, gt_idx = formats.multi_view_confidences(pd, gt, [torch.tensor([0, 1]), torch.tensor([1])])
    assert pd_idx.tolist() == [0, 1] and gt_idx.tolist() == [1, 0]
    np.testing.assert_allclose(conf.numpy(), [0.75, 1.0], rtol=1e-6)


def test_loss_library_matches_reference_g13():
    """a22: the vsrd.losses call surface (pure PyTorch) against the reference's own outputs."""
    from conftest import load_golden
    from vsrd_amd import loss_library as L
    g = load_golden("g13_loss_library")
    p, t = g["p"], g["t"]
    for name in ("cross_entropy", "binary_cross_entropy", "kl_divergence", "binary_kl_divergence", "js_divergence",
                 "binary_js_divergence", "focal_loss", "quality_focal_loss", "tversky_loss", "focal_tversky_loss"):
        torch.testing.assert_close(getattr(L, name)(p, t, reduction="none"), g["out_" + name], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(L.cross_entropy(p, t, dim=1), g["out_cross_entropy_dim1"], rtol=1e-5, atol=1e-6)
    a, b = g["img_a"], g["img_b"]
    torch.testing.assert_close(L.ssim_loss(a, b, reduction="none"), g["out_ssim_loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(L.photometric_loss(a, b, reduction="none"), g["out_photometric_loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(L.smoothness_loss(a[:, :1], b, reduction="none"), g["out_smoothness_loss"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(L.motion_smoothness_loss(a, reduction="none"), g["out_motion_smoothness_loss"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(L.motion_sparsity_loss(a - 0.5, reduction="sum"), g["out_motion_sparsity_loss"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(L.gaussian_nll(g["mean"], g["var"], g["target"], reduction="none"), g["out_gaussian_nll"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(L.student_nll(g["mean"], g["shape"], g["scale"], g["target"], reduction="none"), g["out_student_nll"], rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        L.cross_entropy(p, t, reduction="median")


def test_loss_library_rest_matches_reference_g18():
    """a22, second half: geometric_losses.py and the logit-space NLLs / Monte-Carlo energy scores of probabilistic_losses.py against
    the reference's outputs.  The energy scores sample from torch's global generator: seeding as the fixture generator did replays
    the reference's draws (one rsample([num_samples]) call per score)."""
    from conftest import load_golden
    from vsrd_amd import loss_library as L
    g = load_golden("g18_loss_library_rest")
    torch.testing.assert_close(L.rotation_consistency_loss(g["source"], g["target"], reduction="none"), g["out_rotation_consistency_loss"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(L.translation_consistency_loss(g["source"], g["target"], reduction="none"), g["out_translation_consistency_loss"], rtol=1e-5, atol=1e-7)
    assert float(g["out_rotation_consistency_loss"][0]) < 1e-6 and float(g["out_translation_consistency_loss"][0]) < 1e-6      # target = source^-1
    torch.testing.assert_close(L.sampson_epipolar_distance(g["keypoints_1"], g["keypoints_2"], g["fundamental"][:, None], reduction="none"),
                               g["out_sampson_epipolar_distance"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(L.logit_gaussian_nll(g["mean"], g["var"], g["unit_targets"], reduction="none"), g["out_logit_gaussian_nll"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(L.logit_student_nll(g["mean"], g["shape"], g["scale"], g["unit_targets"], reduction="none"), g["out_logit_student_nll"], rtol=1e-4, atol=1e-5)
    n, seed = int(g["num_samples"]), int(g["seed"])
    for name, args in (("gaussian_energy_score", (g["mean"], g["var"], g["target_values"])),
                       ("student_energy_score", (g["mean"], g["shape"], g["scale"], g["target_values"])),
                       ("logit_gaussian_energy_score", (g["mean"], g["var"], g["unit_targets"])),
                       ("logit_student_energy_score", (g["mean"], g["shape"], g["scale"], g["unit_targets"]))):
        torch.manual_seed(seed)
        torch.testing.assert_close(getattr(L, name)(*args, num_samples=n, reduction="none"), g["out_" + name], rtol=1e-5, atol=1e-6)
    # reductions and the mean default, as the reference's @reduced decorator
    torch.manual_seed(seed)
    mean = L.gaussian_energy_score(g["mean"], g["var"], g["target_values"], num_samples=n)
    torch.testing.assert_close(mean, g["out_gaussian_energy_score"].mean(), rtol=1e-5, atol=1e-6)
    # every public name of the reference's vsrd/losses/*.py is there
    for name in ("cross_entropy binary_cross_entropy kl_divergence binary_kl_divergence js_divergence binary_js_divergence focal_loss quality_focal_loss "
                 "tversky_loss focal_tversky_loss rotation_consistency_loss translation_consistency_loss sampson_epipolar_distance ssim_loss photometric_loss "
                 "gaussian_nll student_nll gaussian_energy_score student_energy_score logit_gaussian_nll logit_student_nll logit_gaussian_energy_score "
                 "logit_student_energy_score smoothness_loss motion_smoothness_loss motion_sparsity_loss").split():
        assert callable(getattr(L, name)), name


def test_capture_gate_keeps_captures_alone():
    """optimization._CaptureGate: a capture runs alone; everything else a frame's thread calls into HIP (construction, eager steps, replay
    launches, graph destruction, host synchronisations) runs next to each other but never during a capture; a waiting capture goes before
    newcomers but does not stop a thread that is already inside from nesting; the garbage collector's non-blocking attempt is refused
    during a capture, inside the capturing thread too."""
    import threading
    import time
    from vsrd_amd import optimization
    gate = optimization._CaptureGate()
    log, inside = [], threading.Event()

    def other(tag, hold, nested=False):
        with gate.replaying():
            log.append(("in", tag))
            inside.set()
            time.sleep(hold)
            if nested:
                with gate:                                # nesting while a capture waits: must not deadlock
                    log.append(("nested", tag))
            log.append(("out", tag))

    first = threading.Thread(target=other, args=("a", 0.3, True))
    second = threading.Thread(target=other, args=("b", 0.0))
    first.start()
    assert inside.wait(5.0)
    second.start()                                        # next to "a": does not wait
    second.join(5.0)
    assert ("out", "b") in log and ("out", "a") not in log
    assert gate.acquire(blocking=False) is True           # ... and so may anybody else, the collector included
    gate.release()
    start = time.perf_counter()
    with gate.capture():                                  # waits for "a", which nests once on its way out
        assert ("nested", "a") in log and ("out", "a") in log and time.perf_counter() - start > 0.1
        assert gate.acquire(blocking=False) is False      # the collector inside the capturing thread: not now
        with gate:                                        # the capturing thread itself passes
            pass
        late = threading.Thread(target=other, args=("c", 0.0))
        late.start()
        time.sleep(0.1)
        assert ("in", "c") not in log                     # everybody else waits for the capture
        refused = []
        probe = threading.Thread(target=lambda: refused.append(gate.acquire(blocking=False)))
        probe.start(); probe.join(5.0)
        assert refused == [False]
    late.join(5.0)
    first.join(5.0)
    assert ("out", "c") in log and gate.capture_seconds > 0.0
    with gate:
        try:
            with gate.capture():                          # a capture inside a held section would wait for itself
                raise AssertionError("unreachable")
        except RuntimeError:
            pass


def test_to_host_rebuilds_containers_and_leaves_the_rest():
    """formats.to_host: what launcher.main hands to torch.save for a frame optimised next to others -- every tensor a host copy, dicts /
    lists / tuples rebuilt with their types, everything else untouched."""
    import collections
    import torch
    from vsrd_amd import formats
    payload = collections.OrderedDict(step=7, models={"m": {"w": torch.arange(4.0).requires_grad_(True)}}, groups=[{"lr": 0.5, "params": (0, 1)}],
                                      metrics={}, name="frame")
    out = formats.to_host(payload)
    assert type(out) is collections.OrderedDict and list(out) == list(payload)
    assert out["step"] == 7 and out["name"] == "frame" and out["groups"] == [{"lr": 0.5, "params": (0, 1)}] and type(out["groups"][0]["params"]) is tuple
    w = out["models"]["m"]["w"]
    assert w.device.type == "cpu" and not w.requires_grad and torch.equal(w, torch.arange(4.0)) and w is not payload["models"]["m"]["w"]


def test_frame_arena_rows_share_one_layout():
    """optimization.FrameArena / FrameRow (frame batches, include/vsrd_hip.h ABI 8): every frame's copy of a buffer sits at the same offset of its
    row, rows are `stride` bytes apart, offsets are 256-byte aligned, views keep dtype and shape and alias the arena, and a row that runs out
    raises instead of walking into its neighbour."""
    from vsrd_amd import optimization
    arena = optimization.FrameArena(3, 10_000, "cpu")
    assert arena.stride % 256 == 0 and arena.stride >= 10_000 and arena.buffer.shape == (3, arena.stride)
    views = []
    for row in arena.rows:
        a = row.new((5, 3), torch.float32, fill=1.5)
        b = row.new(7, torch.int64, fill=2)
        c = row.new((), torch.float32, fill=0.25)
        d = row.adopt(torch.arange(6, dtype=torch.uint8).reshape(2, 3))
        views.append((a, b, c, d))
        assert a.shape == (5, 3) and a.dtype == torch.float32 and b.dtype == torch.int64 and c.shape == () and float(c) == 0.25
        assert torch.equal(d, torch.arange(6, dtype=torch.uint8).reshape(2, 3)) and all(t.is_contiguous() for t in (a, b, c, d))
    assert arena.rows[0].layout == arena.rows[1].layout == arena.rows[2].layout
    assert all(offset % 256 == 0 for offset, _ in arena.rows[0].layout)
    for k in range(4):           # frame f's buffer is exactly f * stride bytes behind frame 0's
        assert views[1][k].data_ptr() - views[0][k].data_ptr() == arena.stride and views[2][k].data_ptr() - views[0][k].data_ptr() == 2 * arena.stride
    views[1][0].fill_(9.0)       # rows do not overlap
    assert float(views[0][0].max()) == 1.5 and float(views[2][0].max()) == 1.5
    with pytest.raises(RuntimeError, match="exhausted"):
        arena.rows[0].new(10_000, torch.float32)


def test_frame_batch_item_slots_and_row_size(lib):
    """FrameBatch.item_slots: the MLP adjoint's work items grow with the batch (about 16384 items per launch); FrameBatch._row_bytes covers what a
    frame of the reference's size allocates (1.4 GB: directions, soft masks, the sampling table, and 0.8 GB of render workspace -- it is sized so that
    every form of the residual step fits, the one-kernel fallback's per-wave caches included)."""
    from vsrd_amd import optimization
    config = optimization.OptimizationConfig()
    assert [optimization.FrameBatch.item_slots(b, config, 8) for b in (1, 2, 4, 8, 16, 64)] == [4, 4, 8, 16, 32, 32]
    assert optimization.FrameBatch.item_slots(8, optimization.OptimizationConfig(num_samples=64), 16) == 16      # two rounds per ray, twice the instances
    row = optimization.FrameBatch._row_bytes((17, 376, 1408, 8), config)
    pixels = 17 * 376 * 1408
    assert row > pixels * (12 + 32 + 4) + lib.vsrd_ray_table_bytes(pixels) + lib.vsrd_residual_step_workspace_bytes(8, 100, 1000) and row < 3 << 29


def test_initial_draw_runs_on_one_thread_and_draws_the_same_values():
    """optimization._InitialDraw: the host-side draw of a frame's initial parameters runs on ONE intra-op thread (256 host cores made it 100 ms) and
    restores the count; the values do not depend on it."""
    from vsrd_amd import models, optimization
    before = torch.get_num_threads()

    def draw():
        torch.default_generator.manual_seed(7)
        return [p.detach().clone() for p in models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).parameters()]

    with optimization._initial_draw:
        assert torch.get_num_threads() == 1
        single = draw()
    assert torch.get_num_threads() == before
    for a, b in zip(single, draw()):
        assert torch.equal(a, b)
