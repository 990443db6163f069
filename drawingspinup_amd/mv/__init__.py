"""Multi-view diffusion (Wonder3D-style 6-view x 2-domain UNet, DDIM, VAE) on gfx950 kernels."""
