class COCOeval:
    pass
