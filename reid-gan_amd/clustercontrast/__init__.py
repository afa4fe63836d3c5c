"""Drop-in `clustercontrast` package (model + trainer side of cluster-contrast-reid-main/clustercontrast) on the
MI355X HIP kernels."""
