"""Host-side mirror of the reference's `model` package for the BMHRL hot path (same class names, constructor
arguments, attribute names and state-dict keys); all arithmetic runs in the HIP kernels."""
