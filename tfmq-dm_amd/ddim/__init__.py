"""Pixel-space DDPM/DDIM side of the hot path: UNet description, schedules and sampler."""
