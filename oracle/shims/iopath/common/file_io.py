"""Shim: g_pathmgr.open == builtin open for local paths."""


class _PathManager:
    def open(self, path, mode="r", **kwargs):
        return open(path, mode)

    def exists(self, path):
        import os
        return os.path.exists(path)

    def isfile(self, path):
        import os
        return os.path.isfile(path)

    def ls(self, path):
        import os
        return os.listdir(path)

    def mkdirs(self, path):
        import os
        os.makedirs(path, exist_ok=True)

    def get_local_path(self, path, **kwargs):
        return path


g_pathmgr = _PathManager()
PathManager = _PathManager
