"""``PNP`` registry + ``build_pnp`` (mirror of /root/reference/monorun/ops/least_squares/builder.py:1-7).

With mmcv importable the real ``mmcv.utils.Registry`` is used, so ``dict(type='PnPUncert', ...)`` in the
reference's configs (configs/kitti_car.py:118-123) resolves exactly as before; without mmcv a minimal
registry with the same two entry points stands in.
"""
try:                                                    # pragma: no cover - mmcv is absent in this image
    from mmcv.utils import Registry, build_from_cfg
except Exception:                                       # noqa: BLE001
    class Registry:
        def __init__(self, name):
            self._name = name
            self._module_dict = {}

        @property
        def name(self):
            return self._name

        @property
        def module_dict(self):
            return self._module_dict

        def get(self, key):
            return self._module_dict.get(key)

        def register_module(self, name=None, force=False, module=None):
            def _register(cls):
                key = name or cls.__name__
                if not force and key in self._module_dict:
                    raise KeyError(f'{key} is already registered in {self._name}')
                self._module_dict[key] = cls
                return cls
            if module is not None:
                return _register(module)
            return _register

    def build_from_cfg(cfg, registry, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise KeyError('cfg must be a dict containing the key "type"')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            obj_cls = registry.get(obj_type)
            if obj_cls is None:
                raise KeyError(f'{obj_type} is not in the {registry.name} registry')
        else:
            obj_cls = obj_type
        return obj_cls(**args)

PNP = Registry('pnp')


def build_pnp(cfg, **default_args):
    return build_from_cfg(cfg, PNP, default_args)
