"""On-disk dataset format of the reference (mrgcn/data/io/)."""
