"""CPU: the C-ABI library loads and exports every symbol include/dream_hip.h declares (no compute), and the
host-side mirror of the reference interface behaves like the reference (validation, resolutions,
state_dict layout, refusal to run without a GPU)."""
import json
import os

import pytest
import torch

import dream_amd
from dream_amd import _hip, image_proc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_library_exports_every_declared_symbol():
    declared = _hip.check_symbols()
    assert len(declared) >= 30 and "dream_conv3x3_nhwc_f32" in declared
    assert set(declared) == set(_hip._SIGNATURES.keys())
    assert _hip.lib().dream_hip_abi_version() == 2
    assert _hip.lib().dream_conv3x3_num_variants() == 11  # selectable; one more (big-patch) variant is automatic


def test_argument_errors_are_reported_not_thrown():
    lib = _hip.lib()
    rc = lib.dream_conv3x3_nhwc_f32(None, None, None, None, 1, 8, 8, 32, 32, 128, 0, None)
    assert rc != 0 and b"null" in lib.dream_hip_last_error()
    rc = lib.dream_maxpool2_nhwc_f32(None, None, 1, 8, 8, 6, None)
    assert rc != 0


def test_host_side_planning_queries_need_no_gpu():
    """Workspace sizes and the 'does this launch carry the bias gradient' query are pure host functions of the shape."""
    lib = _hip.lib()
    assert lib.dream_conv3x3_wgrad_winograd_fuses_bias(64, 64, 64) == 1 and lib.dream_conv3x3_wgrad_winograd_fuses_bias(512, 256, 272) == 1
    assert lib.dream_conv3x3_wgrad_winograd_fuses_bias(64, 48, 48) == 0            # the register-only kernel: no fused bias
    assert lib.dream_conv3x3_wgrad_winograd_fuses_bias(64, 64, 48) == 0            # fewer dy channels than outputs
    small, big = (int(lib.dream_conv3x3_wgrad_winograd_workspace(b, 100, 100, 256, 256)) for b in (2, 128))
    assert small > 0 and big >= small and big % 4 == 0
    assert int(lib.dream_conv3x3_wgrad_winograd_workspace(2, 100, 100, 60, 64)) == 0   # input channels not a multiple of 64
    rc = lib.dream_conv3x3_wgrad_winograd_bias_nhwc_f32(None, None, None, None, None, 1, 8, 8, 64, 64, 64, 0, None)
    assert rc != 0 and b"null" in lib.dream_hip_last_error()


# ---- the reference's own resolution KATs (test/test_image_proc.py:20-91) -----------------------------
def test_shrink_resolution():
    assert image_proc.shrink_resolution((640, 480), (400, 400)) == (533, 400)
    assert image_proc.shrink_resolution((640, 480), (640, 480)) == (640, 480)


def test_shrink_and_crop_resolution():
    assert image_proc.shrink_and_crop_resolution((640, 480), (400, 400)) == ((480, 480), (80, 0))
    assert image_proc.shrink_and_crop_resolution((640, 480), (640, 480)) == ((640, 480), (0, 0))


def test_resolution_after_preprocessing():
    f = image_proc.resolution_after_preprocessing
    assert f((640, 480), (400, 400), "none") == (640, 480)
    assert f((640, 480), (400, 400), "resize") == (400, 400)
    assert f((640, 480), (400, 400), "shrink") == (533, 400)
    assert f((640, 480), (400, 400), "shrink-and-crop") == (400, 400)


@pytest.mark.parametrize("arch,manip,out_res", [("vgg_q", "panda", (100, 100)), ("vgg_f", "panda", (400, 400)),
                                                ("resnet_h", "panda", (208, 208)), ("resnet_f", "baxter", (416, 416))])
def test_network_surface_and_state_dict(arch, manip, out_res, capsys):
    net = dream_amd.create_network_from_config_data(dream_amd.default_network_config(arch, manip))
    assert "DreamNetwork:__init()" in capsys.readouterr().out               # banner (network.py:118-122)
    assert net.trained_net_output_resolution() == out_res
    assert net.network_config["training"]["config"]["net_output_resolution"] == list(out_res)
    assert net.trained_net_input_resolution() == (400, 400)
    assert net.use_belief_peak_scores is True and net.belief_peak_next_best_score == 0.25
    assert net.n_keypoints == len(net.keypoint_names) == len(net.friendly_keypoint_names) == len(net.ros_keypoint_frames)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))[arch]
    sd = net.model.state_dict()
    assert list(sd.keys()) == list(man.keys())                               # module.-prefixed, reference order
    assert all(list(sd[k].shape) == v for k, v in man.items())
    for need in ("train", "loss", "inference", "keypoints_from_image", "enable_training", "enable_evaluation",
                 "net_resolutions_from_image_raw_resolution", "net_output_resolution_from_input_resolution",
                 "save_network_config", "save_network_params", "save_network", "image_preprocessing"):
        assert callable(getattr(net, need))
    assert net.net_resolutions_from_image_raw_resolution((640, 480)) == ((400, 400), out_res)
    assert net.optimizer is None
    with pytest.raises(AssertionError):
        net.train([torch.zeros(1, 3, 8, 8)], torch.zeros(1))                 # "Use enable_training() first."
    net.enable_training()
    assert net.optimizer is not None and net.model.training
    net.enable_evaluation()
    assert not net.model.training


def test_config_validation_uses_assertion_errors():
    good = dream_amd.default_network_config("vgg_q")
    for path in (["architecture"], ["manipulator"], ["training"], ["architecture", "type"],
                 ["architecture", "image_normalization"], ["architecture", "output_heads"],
                 ["training", "config", "net_input_resolution"], ["training", "platform"]):
        cfg = json.loads(json.dumps(good))
        d = cfg
        for k in path[:-1]:
            d = d[k]
        del d[path[-1]]
        with pytest.raises(AssertionError):
            dream_amd.create_network_from_config_data(cfg)
    cfg = json.loads(json.dumps(good))
    cfg["architecture"]["type"] = "transformer"
    with pytest.raises(AssertionError):
        dream_amd.create_network_from_config_data(cfg)
    cfg = json.loads(json.dumps(good))
    cfg["training"]["config"]["net_output_resolution"] = [50, 50]           # disagrees with the model
    with pytest.raises(AssertionError):
        dream_amd.create_network_from_config_data(cfg)
    assert dream_amd.KNOWN_OPTIMIZERS == ["adam", "sgd"] and dream_amd.KNOWN_ARCHITECTURES == ["vgg", "resnet"]


def test_save_and_reload_roundtrip(tmp_path):
    net = dream_amd.create_network_from_config_data(dream_amd.default_network_config("vgg_q"))
    out = tmp_path / "run"
    net.save_network(str(out), "epoch_1")
    with pytest.raises(AssertionError):
        net.save_network(str(out), "epoch_2")                                # directory exists, overwrite=False
    with pytest.raises(AssertionError):
        net.save_network_params(str(out / "epoch_1.pth"))                    # file exists, overwrite=False
    net.save_network(str(out), "epoch_1", overwrite=True)
    again = dream_amd.create_network_from_config_file(str(out / "epoch_1.yaml"), str(out / "epoch_1.pth"))
    for (k1, v1), (k2, v2) in zip(net.model.state_dict().items(), again.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_reads_reference_style_omap_yaml(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("!!omap\n- architecture: !!omap\n  - type: vgg\n  - input_heads:\n    - image_rgb\n- n: 3\n")
    from dream_amd.network import _load_yaml
    assert _load_yaml(str(p)) == {"architecture": {"type": "vgg", "input_heads": ["image_rgb"]}, "n": 3}


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = dream_amd.create_network_from_config_data(dream_amd.default_network_config("vgg_q"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.inference(torch.zeros(1, 3, 400, 400))
    with pytest.raises(RuntimeError):
        dream_amd.peaks_from_belief_maps(torch.zeros(1, 8, 8), 0.0)


def test_python_surface_matches_reference_signatures():
    """The drop-in boundary: every function / method / constructor parameter a caller of the reference can
    name exists here with the same name, order and default (tests/golden/api_surface.json, read off the reference with
    inspect by make_golden.py --only-api), and a constructed DreamNetwork carries the reference's public attributes."""
    import inspect
    import json
    import dream_amd
    from dream_amd import image_proc, models, network, spatial_softmax
    surface = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_surface.json")))

    def params(fn):
        return [[n, None if q.default is inspect._empty else repr(q.default)]
                for n, q in inspect.signature(fn).parameters.items()]

    ref = surface["dream.network"]
    for name, sig in ref["functions"].items():
        assert params(getattr(network, name)) == sig, name
        assert getattr(dream_amd, name) is getattr(network, name)              # dream/__init__.py re-exports them
    assert list(network.KNOWN_ARCHITECTURES) == ref["constants"]["KNOWN_ARCHITECTURES"]
    assert list(network.KNOWN_OPTIMIZERS) == ref["constants"]["KNOWN_OPTIMIZERS"]
    for name, sig in ref["DreamNetwork"].items():
        assert params(getattr(network.DreamNetwork, name)) == sig, name
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = dream_amd.create_network_from_config_data(dream_amd.default_network_config("vgg_q"))
    assert set(ref["DreamNetwork.instance_attributes"]) <= set(vars(net))
    for mod, key in ((models, "dream.models"), (spatial_softmax, "dream.spatial_softmax")):
        for cls, meths in surface[key].items():
            for name, sig in meths.items():
                assert params(getattr(getattr(mod, cls), name)) == sig, (cls, name)
    for name, sig in surface["dream.image_proc"].items():
        assert params(getattr(image_proc, name)) == sig, name


def test_bench_leaves_one_json_error_line_without_a_gpu():
    """bench.py on a box it cannot run on (here: no GPU) must leave ONE JSON line with an `error` field and a non-zero exit status
    -- what makes a failed SCALE record diagnosable -- not a bare traceback."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the error path of this test needs none")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=root, capture_output=True,
                         text=True, timeout=300)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode != 0 and len(lines) == 1, (out.returncode, out.stdout[-500:])
    line = json.loads(lines[0])
    assert line["value"] is None and "needs a GPU" in line["error"] and line["n_gpus"] == 1


def test_deferred_side_hands_the_leaves_over_in_segments(monkeypatch):
    """models._DeferredSide (a backward captured as a sequence of graphs, DREAM_TRAIN_GRAPH_SPLIT): leaves are not run where they are
    queued but handed to the capture controller every `leaves` of them, in order, and the rest at join(); their inputs stay
    referenced until join()."""
    from dream_amd import models

    class Ctl:
        leaves = 3

        def __init__(self):
            self.cuts = []

        def cut(self, fns, join=False):
            self.cuts.append(([fn() for fn in fns], join))

    monkeypatch.setattr(models._SideStream._counts, "leaves", 0, raising=False)       # number of leaves unknown: equal segments
    ctl = Ctl()
    side = models._DeferredSide(ctl)
    ran = []
    for i in range(7):
        side.run(lambda i=i: ran.append(i) or i, "in%d" % i)
        assert len(ran) == 3 * ((i + 1) // 3) and len(side.keep) == i + 1
    side.join()
    assert ctl.cuts == [([0, 1, 2], False), ([3, 4, 5], False), ([6], True)] and side.keep == [] and side.pending == []
    side.join()                                            # nothing pending: still a cut, so that the main stream waits
    assert ctl.cuts[-1] == ([], True)
    # number of leaves known from the eager step before: the last segments shrink to single leaves, one leaf is left for join()
    monkeypatch.setattr(models._SideStream._counts, "leaves", 20, raising=False)
    ctl = Ctl()
    ctl.leaves = 8
    side = models._DeferredSide(ctl)
    for i in range(20):
        side.run(lambda i=i: i)
    side.join()
    assert [len(c[0]) for c in ctl.cuts] == [8, 6, 3, 1, 1, 1] and [c[0] for c in ctl.cuts][-1] == [19] and ctl.cuts[-1][1], ctl.cuts
