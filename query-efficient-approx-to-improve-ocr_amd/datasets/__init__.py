"""Datasets: synthetic generators and the reference on-disk formats (ImgDataset, PatchDataset)."""
