# `data` package of the HIP drop-in (device-side input pipeline only)
