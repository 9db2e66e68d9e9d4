class VisionDataset:
    def __init__(self, *a, **k):
        pass
