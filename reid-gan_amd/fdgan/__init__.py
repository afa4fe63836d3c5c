"""Drop-in `fdgan` package (FD-GAN-master/fdgan): networks, losses and the FDGANModel step driver on the
MI355X HIP kernels."""
