"""Name -> class registry: the reference's plugin API (lavis/common/registry.py:9-329) — same method names, same
duplicate-registration errors, same ``get`` with dotted paths."""


class Registry:
    mapping = {k: {} for k in ("builder", "task", "model", "processor", "lr_scheduler", "runner", "state", "paths")}

    @classmethod
    def _register(cls, kind, name, base=None, base_name=""):
        def wrap(obj):
            if base is not None:
                assert issubclass(obj, base), f"All {kind}s must inherit {base_name} class"
            if name in cls.mapping[kind]:
                raise KeyError(f"Name '{name}' already registered for {cls.mapping[kind][name]}.")
            cls.mapping[kind][name] = obj
            return obj

        return wrap

    @classmethod
    def register_builder(cls, name):
        return cls._register("builder", name)

    @classmethod
    def register_task(cls, name):
        return cls._register("task", name)

    @classmethod
    def register_model(cls, name):
        from lavis.models.base_model import BaseModel

        return cls._register("model", name, BaseModel, "BaseModel")

    @classmethod
    def register_processor(cls, name):
        return cls._register("processor", name)

    @classmethod
    def register_lr_scheduler(cls, name):
        return cls._register("lr_scheduler", name)

    @classmethod
    def register_runner(cls, name):
        return cls._register("runner", name)

    @classmethod
    def register_path(cls, name, path):
        assert isinstance(path, str), "All path must be str."
        if name in cls.mapping["paths"]:
            raise KeyError(f"Name '{name}' already registered.")
        cls.mapping["paths"][name] = path

    @classmethod
    def register(cls, name, obj):
        path = name.split(".")
        cur = cls.mapping["state"]
        for part in path[:-1]:
            cur = cur.setdefault(part, {})
        cur[path[-1]] = obj

    @classmethod
    def get_builder_class(cls, name):
        return cls.mapping["builder"].get(name)

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model"].get(name)

    @classmethod
    def get_task_class(cls, name):
        return cls.mapping["task"].get(name)

    @classmethod
    def get_processor_class(cls, name):
        return cls.mapping["processor"].get(name)

    @classmethod
    def get_lr_scheduler_class(cls, name):
        return cls.mapping["lr_scheduler"].get(name)

    @classmethod
    def get_runner_class(cls, name):
        return cls.mapping["runner"].get(name)

    @classmethod
    def list_runners(cls):
        return sorted(cls.mapping["runner"])

    @classmethod
    def list_models(cls):
        return sorted(cls.mapping["model"])

    @classmethod
    def list_tasks(cls):
        return sorted(cls.mapping["task"])

    @classmethod
    def list_processors(cls):
        return sorted(cls.mapping["processor"])

    @classmethod
    def list_lr_schedulers(cls):
        return sorted(cls.mapping["lr_scheduler"])

    @classmethod
    def list_datasets(cls):
        return sorted(cls.mapping["builder"])

    @classmethod
    def get_path(cls, name):
        return cls.mapping["paths"].get(name)

    @classmethod
    def get(cls, name, default=None, no_warning=False):
        value = cls.mapping["state"]
        for part in name.split("."):
            value = value.get(part, default) if isinstance(value, dict) else default
            if value is default:
                break
        return value

    @classmethod
    def unregister(cls, name):
        return cls.mapping["state"].pop(name, None)


registry = Registry()
