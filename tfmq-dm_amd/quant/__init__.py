"""Mirror of the reference's `quant/` surface (QuantModel, QuantLayer, quant blocks, calibration,
reconstruction, adaptive rounding) executing on the MI355X HIP kernels.  Put the directory
`tfmq-dm_amd/` on sys.path to use it as a drop-in `quant` package."""
