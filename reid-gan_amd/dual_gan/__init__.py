"""Drop-in `dual_gan` package (model side of cluster-contrast-reid-main/dual_gan) on the MI355X HIP kernels."""
