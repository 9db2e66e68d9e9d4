"""Shim: placeholder so train/data/sam3_image_dataset.py finds a 'video backend'."""
