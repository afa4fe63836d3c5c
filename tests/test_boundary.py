"""CPU: the C-ABI boundary — the shared library loads, exports every symbol include/reidgan_hip.h declares,
rejects bad arguments with a message, and the Python host refuses to run without a GPU (no fallback)."""
import ctypes
import os
import subprocess

import pytest
import torch

from rg_hip import lib as L
from rg_hip import ops


def test_header_parses_and_library_exports_every_symbol():
    protos = L.parse_header()
    assert len(protos) >= 45
    assert os.path.exists(L.LIB_PATH), "build with __graft_entry__.build() first"
    dll = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, missing
    # and nothing exported by the library is undocumented in the header
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], stdout=subprocess.PIPE,
                         universal_newlines=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T rg_" in ln}
    assert exported <= set(protos), exported - set(protos)


def test_queries_and_error_convention_without_gpu():
    lib = L.lib
    assert lib.rg_version() >= 1
    assert lib.rg_family_count() == len(L.FAMILIES)
    assert lib.rg_loss_workspace() >= 1024
    assert lib.rg_conv2d_wgrad_workspace(32, 64, 64, 3, 3, 64, 32) > 0
    # every host-only query of the header answers on a machine without a GPU; queries return VALUES (size_t), never a status —
    # an `int` prototype would be read as an error code by the binding
    protos = L.parse_header()
    queries = {"rg_conv2d_fwd_workspace": (32, 64, 64, 3, 3, 64, 32), "rg_conv2d_dgrad_workspace": (32, 64, 64, 32, 64, 3, 3, 1, 1),
               "rg_conv2d_wgrad_workspace": (32, 64, 64, 3, 3, 64, 32), "rg_bn_workspace": (32, 64, 2048),
               "rg_bn_train_fused_ok": (64, 1024, 128), "rg_conv2d_f8_wgrad_workspace": (32, 64, 64, 3, 3, 64, 32),
               "rg_spectral_norm_bwd_workspace": (256, 2304), "rg_loss_workspace": (), "rg_channel_sum_ok": (64, 256, 128)}
    assert sorted(queries) == sorted(n for n, (ret, _) in protos.items() if ret == "size_t")
    for name, a in queries.items():
        assert getattr(lib, name)(*a) >= 0, name
    assert lib.rg_bn_train_fused_ok(64, 1024, 128) == 1 and lib.rg_bn_train_fused_ok(64, 64, 2048) == 0
    assert lib.rg_channel_sum_ok(64, 256, 128) == 1 and lib.rg_channel_sum_ok(32, 64, 2048) == 0
    assert lib.rg_f8_grad_tiles(17, 65) == 4        # a plain count, not a status: 2 sample tiles x 2 pixel tiles
    with pytest.raises(RuntimeError) as e:
        lib.rg_fill(None, 4, 1.0, None)
    assert "rg_fill" in str(e.value)
    with pytest.raises(RuntimeError) as e:       # inconsistent geometry is rejected before any launch
        lib.rg_conv2d_fwd(1, 1, None, 1, 1, 3, 8, 8, 4, 3, 3, 1, 1, 1, 1, 99, 99, None, None, None, 0, 0.0, None, 0, None)
    assert "rg_conv2d_fwd" in str(e.value)


def test_host_refuses_cpu_tensors():
    x = torch.randn(1, 3, 8, 8)
    w = torch.randn(4, 3, 3, 3)
    with pytest.raises(RuntimeError) as e:
        ops.conv2d_fwd(x, w, 1, 1)
    assert "no CPU fallback" in str(e.value)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    fresh = L._Lib()
    with pytest.raises(ImportError):
        fresh.load()


def test_state_dict_layout_matches_reference_keys():
    """Checkpoint compatibility: parameter names/shapes equal the oracle's (== the reference's) modules."""
    from oracle import ref_torch as O
    import fdgan.networks as N
    import reid.models as RM
    from reid.models.embedding import EltwiseSubEmbed
    from reid.models.multi_branch import SiameseNet

    def same(a, b):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        for k in sa:
            assert sa[k].shape == sb[k].shape, k

    same(N.CustomPoseGenerator(128, 2048, 256, dropout=0.2, norm_layer=N.get_norm_layer('batch')),
         O.OPoseGenerator(128, 2048, 256, dropout=0.2))
    same(N.CustomPoseGenerator(128, 2048, 256, norm_layer=N.get_norm_layer('batch'), connect_layers=3),
         O.OPoseGenerator(128, 2048, 256, connect_layers=3))
    same(N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer('batch')), O.OPatchDiscriminator(21))
    same(N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer('instance')), O.OPatchDiscriminator(21, 'instance'))
    e = SiameseNet(RM.create('resnet50', pretrained=False, cut_at_pooling=True),
                   EltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048, num_classes=2))
    oe = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True),
                       O.OEltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048, num_classes=2))
    same(e, oe)
    assert "base_model.base.fc.weight" in e.state_dict()       # the unused fc stays in checkpoints
    with pytest.raises(KeyError):
        RM.create('resnet51')


def test_dualgan_state_dict_keys_match_reference_layout():
    """the oracle's keys ARE the reference's (make_golden_dualgan loads one into the other)"""
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    og, _ = C.posegen1_case()
    rg = N.PoseGenerator1(64, 18, 256, 3, 'instance', 'LeakyReLU', False, False, 3, True, 2, 2, 2)
    assert list(rg.state_dict().keys()) == list(og.state_dict().keys())
    od, _ = C.resdisc_case()
    rd = N.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    assert sorted(rd.state_dict().keys()) == sorted(od.state_dict().keys())
    for k, v in od.state_dict().items():
        assert rd.state_dict()[k].shape == v.shape, k


def test_dualgan_api_errors():
    """error behaviour of the dual_gan factories (CC/dual_gan/models/networks.py:14-33, base_function.py:38-63)"""
    import argparse
    from dual_gan.models import base_function as BF
    from dual_gan.models import networks as N
    from dual_gan.models.external_function import GANLoss
    with pytest.raises(NotImplementedError):
        BF.get_norm_layer('group')
    with pytest.raises(NotImplementedError):
        BF.get_nonlinearity_layer('Swish')
    with pytest.raises(NotImplementedError):
        GANLoss('nonsense')
    with pytest.raises(TypeError):
        N.define_G(argparse.Namespace(model_gen='nope', init_type='orthogonal'), 3, 18)
    assert BF.get_norm_layer('none') is None


def test_average_meter_interface():
    """attribute interface the trainers' progress lines read (CC/clustercontrast/utils/meters.py)"""
    from clustercontrast.utils.meters import AverageMeter
    m = AverageMeter()
    assert (m.val, m.avg, m.sum, m.count) == (0, 0, 0, 0)
    m.update(2.0)
    m.update(4.0, n=3)
    assert m.val == 4.0 and m.sum == 14.0 and m.count == 4 and m.avg == 3.5
    m.reset()
    assert (m.val, m.avg, m.sum, m.count) == (0, 0, 0, 0)


def test_dual_gan_registry():
    import dual_gan.models as DM
    from dual_gan.models.AE_model import AEModel
    assert DM.find_model_using_name("AE") is AEModel
    assert DM.get_option_setter("AE") == AEModel.modify_options
    import pytest
    with pytest.raises(ImportError):
        DM.find_model_using_name("nope")


def test_generated_cpython_binding_is_opt_in_and_hash_checked():
    """lib/_rg_native*.so (generated from the header by csrc/gen_pymod.py; default binding, RG_NATIVE_BIND=0 selects ctypes) serves
    every entry point with the ctypes binding's conventions — values for queries, RuntimeError + rg_last_error() text for a
    non-zero status, TypeError for a wrong argument count — and carries the hash of the prototypes it was generated from: a
    module whose hash differs from the parsed header is refused (bind() matches symbols by name only)."""
    import subprocess
    import sys
    lib = L.lib
    lib.load()
    if os.environ.get("RG_NATIVE_BIND", "1") != "0":
        assert lib._native is not None, "the generated binding was not built (make -C reid-gan_amd/csrc)"
    want = [lib.rg_version(), lib.rg_bn_train_fused_ok(64, 1024, 128), lib.rg_conv2d_wgrad_workspace(32, 64, 64, 3, 3, 64, 32)]
    with pytest.raises(RuntimeError, match="rg_fill failed"):
        lib.rg_fill(None, 4, 1.0, None)
    code = ("import sys, warnings; sys.path.insert(0, %r); import rg_hip.lib as L\n"
            "%s"
            "lib = L.lib; lib.load(); print(int(lib._native is not None))\n"
            "print(lib.rg_version(), lib.rg_bn_train_fused_ok(64, 1024, 128), lib.rg_conv2d_wgrad_workspace(32, 64, 64, 3, 3, 64, 32))\n"
            "try:\n    lib.rg_fill(None, 4, 1.0, None)\nexcept RuntimeError as e:\n    print(int('rg_fill failed' in str(e)))\n"
            "try:\n    lib.rg_fill(None, 4)\nexcept TypeError:\n    print(1)\n")

    def run(pre):
        env = dict(os.environ, RG_NATIVE_BIND="1", PYTHONWARNINGS="ignore")
        out = subprocess.run([sys.executable, "-c", code % (L.PKG_ROOT, pre)], env=env, stdout=subprocess.PIPE,
                             universal_newlines=True, check=True).stdout.split()
        return [int(v) for v in out]
    assert run("") == [1] + want + [1, 1]                                            # the generated binding, same answers
    env0 = dict(os.environ, RG_NATIVE_BIND="0", PYTHONWARNINGS="ignore")
    out0 = subprocess.run([sys.executable, "-c", code % (L.PKG_ROOT, "")], env=env0, stdout=subprocess.PIPE, universal_newlines=True,
                          check=True).stdout.split()
    assert [int(v) for v in out0] == [0] + want + [1, 1]                             # ctypes, same answers
    # a module generated from other prototypes: refused, calls go through ctypes (same answers)
    assert run("L.proto_hash = lambda protos: 'not-the-hash'\n") == [0] + want + [1, 1]
