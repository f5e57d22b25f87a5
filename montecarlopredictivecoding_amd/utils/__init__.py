"""Counterparts of the reference's ``utils`` helpers that sit on the hot path."""
