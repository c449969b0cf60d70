try:
    from tqdm import tqdm
except ImportError:        # progress bars are cosmetic
    def tqdm(iterable=None, *args, **kwargs):
        class _Bar:
            def __init__(self, it): self.it = it
            def __iter__(self): return iter(self.it)
            def set_postfix(self, **kw): pass
        return _Bar(iterable)
