"""Latent-diffusion side of the hot path: schedules and the DDIM sampler with classifier-free guidance."""
