"""Every __global__ kernel of reid-gan_amd/csrc/*.hip has a row in tools/kernel_families.py (the table behind the per-family HBM
traffic of `roofline.traffic`), so a new kernel cannot silently be booked under "other"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_every_source_kernel_has_a_family():
    import kernel_families as KF
    src = KF.source_kernels(os.path.join(ROOT, "reid-gan_amd", "csrc"))
    assert len(src) > 90
    missing = sorted(k for k in src if k not in KF.FAMILY)
    assert not missing, "kernels without a family row in tools/kernel_families.py: %s" % missing


def test_rocprof_names_resolve():
    import kernel_families as KF
    assert KF.family_of('"conv3x3_halo_kernel<128, false, 13>"') == "conv"
    assert KF.family_of("void (anonymous namespace)::conv_fwd_kernel<128, 128, 2, 2, 1, true>(ConvP)") == "conv"
    assert KF.family_of("bn_fold_wgrad_kernel") == "conv"
    assert KF.family_of("conv_dgrad_smallc_px_kernel<4>") == "conv"
    assert KF.family_of("f8_quantize_dual_kernel") == "conv_f8"
    assert KF.family_of("bn_train_fwd_fused_kernel") == "norm"
    assert KF.family_of("at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float> >") == "other"
    assert KF.family_of("__amd_rocclr_copyBuffer") == "other"
