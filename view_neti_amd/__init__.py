"""view_neti_amd — MI355X-native engine for the ViewNeTI textual-inversion train step."""
__version__ = "0.1.0"
