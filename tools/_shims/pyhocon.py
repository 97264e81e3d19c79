"""Stand-in for pyhocon: dict-backed config with the accessor methods the reference calls."""


class ConfigTree(dict):
    def _get(self, k, d):
        return self[k] if k in self else d

    def get_int(self, k, d=None):
        return int(self._get(k, d))

    def get_float(self, k, d=None):
        return float(self._get(k, d))

    def get_bool(self, k, d=None):
        return bool(self._get(k, d))

    def get_string(self, k, d=None):
        return str(self._get(k, d))

    def get_list(self, k, d=None):
        return self._get(k, d)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return ConfigTree(v) if isinstance(v, dict) and not isinstance(v, ConfigTree) else v


class ConfigFactory:
    @staticmethod
    def from_dict(d):
        return ConfigTree(d)
