backend = None
