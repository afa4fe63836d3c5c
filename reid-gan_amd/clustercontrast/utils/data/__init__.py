"""Device-side pieces of cluster-contrast-reid-main/clustercontrast/utils/data that sit on the training step."""
