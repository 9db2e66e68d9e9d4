class COCO:
    pass
