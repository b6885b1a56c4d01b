"""Drop-in pieces of the reference's `utils` package that sit directly on the train-step path.
Only `utils.metrics.VQAAccuracy` is provided; everything else in the reference's `utils` is out of scope."""
