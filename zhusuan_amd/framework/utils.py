"""Context stack (reference zhusuan/framework/utils.py:20-46)."""

__all__ = ['Context']


class Context(object):
    def __enter__(self):
        type(self).get_contexts().append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        type(self).get_contexts().pop()

    @classmethod
    def get_contexts(cls):
        if '_contexts' not in cls.__dict__:
            cls._contexts = []
        return cls._contexts

    @classmethod
    def get_context(cls):
        try:
            return cls.get_contexts()[-1]
        except IndexError:
            raise RuntimeError("No contexts on the stack.")
