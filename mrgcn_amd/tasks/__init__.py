"""Task-side pieces that sit directly on the encoder's output (reference: mrgcn/tasks/)."""
