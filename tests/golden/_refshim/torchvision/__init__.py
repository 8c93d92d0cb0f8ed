from . import transforms
