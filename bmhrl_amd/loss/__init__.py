"""Host-side mirror of the reference's `loss` package (label smoothing KL, biased KL, REINFORCE)."""
