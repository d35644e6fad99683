class GlobalHydra:
    _inst = None

    @classmethod
    def instance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def is_initialized(self):
        return False

    def clear(self):
        return None
