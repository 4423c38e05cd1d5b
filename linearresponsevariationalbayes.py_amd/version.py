"""Package version (the reference keeps it in LinearResponseVariationalBayes/version.py and imports it from there)."""
__version__ = '0.1.0'
