"""`lib.data_utils` of the reference; falls through to a reference checkout later on sys.path (see lib/__init__.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
